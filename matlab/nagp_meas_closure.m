function c = nagp_meas_closure(H, h)
% NAGP_MEAS_CLOSURE - what the HIP kernels need to know about the EKF measurement handles of the reference drivers
%
%   c = nagp_meas_closure(dhandle, handle)
%
% gf_giekf_modulator_nmf.m:108-113 / gf_giekf_modulator_nmf_constraints.m:136-142 build
%   handle  = @(x,p) funh(x,H,linkf,D,N,Wnmf)          h(x) = (H_z x)' * Wnmf * linkf(H_g x)
%   dhandle = @(x,p) funhd(x,H,linkf,dlinkf,D,N,Wnmf)  its Jacobian
% and pass them to ekf_update1 / iekf_update1.  A MATLAB closure cannot run on the GPU; the captured variables
% (functions(handle).workspace{1}: H, D, N, Wnmf, linkf) say everything the library needs -- every row of H has one
% non-zero (the first state of a block, a power of two after `balance`), the link is the softplus of :103 / :138.
% Any other handle is an error (there is no host-callback path).
%
% Out: struct with h_col (int32, 0-based), h_val, Wnmf, D, N

  info = functions(h);
  if ~isfield(info,'workspace') || isempty(info.workspace), error('nagp:closure','h must be the anonymous function the drivers build around funh'); end
  ws = info.workspace{1};
  need = {'H','D','N','Wnmf','linkf'};
  for i = 1:numel(need)
    if ~isfield(ws,need{i}), error('nagp:closure','the workspace of h holds no %s',need{i}); end
  end
  if isempty(regexp(func2str(h),'funh\(','once')) || (isa(H,'function_handle') && isempty(regexp(func2str(H),'funhd\(','once')))
    error('nagp:closure','h / H must be the funh / funhd closures of gf_giekf_modulator_nmf*.m');
  end
  s = regexprep(func2str(ws.linkf),'\s','');
  if isempty(regexp(s,'^@\((\w+)\)log\(1\+exp\(\1\)\)$','once')), error('nagp:closure','link %s has no GPU implementation in the EKF update',s); end
  [M,S] = size(ws.H);
  if M ~= ws.D + ws.N, error('nagp:closure','H must have D+N rows'); end
  cols = zeros(M,1); vals = zeros(M,1);
  for n = 1:M
    cidx = find(ws.H(n,:));
    if numel(cidx) ~= 1, error('nagp:closure','row %d of H must have exactly one non-zero',n); end
    cols(n) = cidx; vals(n) = ws.H(n,cidx);
  end
  c.h_col = int32(cols - 1); c.h_val = vals; c.Wnmf = ws.Wnmf; c.D = ws.D; c.N = ws.N; c.S = S;
end
