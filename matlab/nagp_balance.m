function [F, L, H, Pinf] = nagp_balance(F, L, H, Pinf)
% NAGP_BALANCE - the balancing step the reference applies before discretising
% (gf_ep_modulator_nmf_constraints.m:115-121, ihgp_ep_modulator_nmf.m:81-87, gf_giekf_modulator_nmf.m:78-84):
% a diagonal similarity of powers of two, so H keeps one non-zero per row (no longer 1).
  [T,F] = balance(F);
  L = T\L; H = H*T;
  LL = T\chol(Pinf,'lower'); Pinf = LL*LL';
end
