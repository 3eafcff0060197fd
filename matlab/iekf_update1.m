function [M,P,K,MU,S,LH] = iekf_update1(M,P,y,H,R,h,V,param,iters)
% IEKF_UPDATE1 - the "iterated" EKF measurement update of the reference ON THE GPU
%
% Same call as the reference's matlab/iekf_update1.m:48 for the handles the drivers pass
% (H = dhandle, h = handle of gf_giekf_modulator_nmf.m:108-113; scalar y, V = [] as on the hot path, :161):
%   for it = 1:iters: H_ = H(M); MU = h(M); S = R + H_*P*H_'; K = P*H_'/S; M = M + K*(y-MU); end; P = P - K*S*K'
% (:110-117 -- not the textbook IEKF, reproduced as written).  LH (the likelihood, :119-121) is evaluated here from
% the returned MU and S exactly as the reference does (gauss_pdf of a scalar).

  if nargin < 5, error('Too few arguments'); end
  if nargin < 7, V = []; end
  if nargin < 9 || isempty(iters), iters = 5; end
  if ~isempty(V), error('nagp:iekf','a noise Jacobian V is never passed on the hot path and is not served'); end
  if ~isa(H,'function_handle') || ~isa(h,'function_handle')
    error('nagp:iekf','H and h must be the dhandle / handle closures of gf_giekf_modulator_nmf*.m (numeric H: use the reference file)');
  end
  if numel(y) ~= 1, error('nagp:iekf','scalar measurements only'); end
  c = nagp_meas_closure(H,h);
  [M,P,K,MU,S] = nagp_mex('iekf_update1', M(:), P, y, c.h_col, c.h_val, c.Wnmf, R, iters);
  if nargout > 5
    LH = exp(-0.5*(y-MU)^2/S) / sqrt(2*pi*S);        % gauss_pdf(y,MU,S)
  end
end
