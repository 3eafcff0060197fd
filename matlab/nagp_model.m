function [model, A, Q] = nagp_model(F, L, Qc, H, Pinf, Wnmf, D, N, lik_param, symmetrise_Q, stationary_Q)
% NAGP_MODEL - discrete-time model struct for nagp_mex from what an `ss` handle returned
%
%   [model, A, Q] = nagp_model(F,L,Qc,H,Pinf,Wnmf,D,N,lik_param [,symmetrise_Q [,stationary_Q]])
%
% Does what the reference functions do right before their loops: [A,Q] = lti_disc(F,L,Qc,1)
% (gf_ep_modulator_nmf.m:108), optionally Q = (Q+Q')/2 (ihgp_ep_modulator_nmf.m:97) or Q = Pinf - A*Pinf*A'
% (gf_giekf_modulator_nmf_constraints.m:378), and describes H as the library wants it: row n of H has ONE non-zero,
% at the first state of block n (ss_modulators_nmf.m:64-78; a power of two after `balance`).

  if nargin < 10, symmetrise_Q = false; end
  if nargin < 11, stationary_Q = false; end
  if stationary_Q
    A = expm(F); Q = Pinf - A*Pinf*A';
  else
    [A,Q] = lti_disc(F,L,Qc,1);
  end
  if symmetrise_Q, Q = (Q+Q')/2; end
  [M,S] = size(H);
  cols = zeros(M,1); hval = zeros(M,1);
  for n = 1:M
    c = find(H(n,:));
    if numel(c) ~= 1, error('nagp:model','row %d of H must have exactly one non-zero',n); end
    cols(n) = c; hval(n) = H(n,c);
  end
  if any(diff(cols) <= 0), error('nagp:model','the blocks of H must be in ascending order'); end
  model.A = A; model.Q = Q; model.Pinf = Pinf;
  model.block_offsets = int32([cols'-1, S]);
  model.h_val = hval;
  model.Wnmf = Wnmf;
  model.D = D; model.N = N;
  model.lik_param = lik_param(1);
end
