function c = nagp_closure(mom, cub_dim)
% NAGP_CLOSURE - what the HIP kernels need to know about a `mom` handle of the reference drivers
%
%   c = nagp_closure(mom, cub_dim)
%
% The moment callbacks the reference passes around are closures over (likfunc, link, p_cubature) or
% (likfunc, link, wn, xn_unscaled) -- demo_toy_modulators_nmf.m:81, demo_toy_modulators.m:81,
% experiments/train_GTFNMF.m:149.  A MATLAB closure cannot run on the GPU, so the wrappers read the captured
% variables and hand the library an enum for the likelihood, an enum + shift for the link and the unit sigma
% points.  Unknown likelihoods / links are an error (there is no host-callback path).
%
% Out: struct with lik_kind (0 likModulatorPower, 1 likModulatorNMFPower, 2 likModulatorPreCalcwn),
%      link_kind (0 log(1+exp(g-shift)), 1 exp(g)), link_shift, wn (1 x n_pts), xn_unscaled (cub_dim x n_pts)

  info = functions(mom);
  if ~isfield(info,'workspace') || isempty(info.workspace)
    error('nagp:closure','mom must be an anonymous function that captured likfunc and link');
  end
  ws = info.workspace{1};
  if ~isfield(ws,'likfunc') || ~isfield(ws,'link')
    error('nagp:closure','the workspace of mom holds no likfunc / link');
  end

  % --- likelihood
  name = ws.likfunc; if isa(name,'function_handle'), name = func2str(name); end
  name = regexprep(char(name),'^@','');
  switch name
    case 'likModulatorPower',     c.lik_kind = 0;
    case 'likModulatorNMFPower',  c.lik_kind = 1;
    case 'likModulatorPreCalcwn', c.lik_kind = 2;
    otherwise, error('nagp:closure','likelihood %s has no GPU implementation',name);
  end

  % --- link: log(1+exp(g)), log(1+exp(g-1)), log(1+exp(g-<captured variable>)), exp(g)
  [c.link_kind,c.link_shift] = nagp_link(ws.link);

  % --- sigma points (unit scale; the kernels apply mean and standard deviation)
  if isfield(ws,'wn') && isfield(ws,'xn_unscaled')
    c.wn = ws.wn(:)'; c.xn_unscaled = ws.xn_unscaled;
    if size(c.xn_unscaled,1) ~= cub_dim, c.xn_unscaled = c.xn_unscaled'; end
  elseif isfield(ws,'p_cubature')
    p = ws.p_cubature;
    if ismember(p,[3,5,7,9])
      [wn,xn] = utp_ws(p,cub_dim);             % symmetric-cubature-rules/utp_ws.m
      c.wn = wn(:)'; c.xn_unscaled = xn;
    else
      [xn,wn] = mvhermgauss(zeros(cub_dim,1),ones(cub_dim,1),p);   % Gauss-Hermite grid on the unit Gaussian
      c.wn = wn(:)'; c.xn_unscaled = xn';
    end
  else
    error('nagp:closure','the workspace of mom holds neither p_cubature nor wn / xn_unscaled');
  end
  if size(c.xn_unscaled,1) ~= cub_dim || size(c.xn_unscaled,2) ~= numel(c.wn)
    error('nagp:closure','sigma points must be cub_dim x n_pts');
  end
end
