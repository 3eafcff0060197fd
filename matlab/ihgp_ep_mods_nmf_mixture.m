function [varargout] = ihgp_ep_mods_nmf_mixture(w,x,y,ss,mom,xt,kernel1,kernel2,J,ep_fraction,ep_damping,ep_itts)
% IHGP_EP_MODS_NMF_MIXTURE - source separation on the infinite-horizon filter / smoother ON THE GPU
%
% Same call as the reference's matlab/experiments/ihgp_ep_mods_nmf_mixture.m:1 (the inference of source_sep_piano.m:137-141).
% Set-up as in the .m (no balancing, Q symmetrised, DARE look-up tables :127-229 through nagp_ihgp_tables); the sweeps
% :248-536 run in libnagp.so under the mixture EP rule (flag 8; R starts at 0 and no abs(Varft): flag 1).

  if nargin < 10, ep_fraction = 0.5; end
  if nargin < 11, ep_damping = 0.1; end
  if nargin < 12, ep_itts = 30; end
  if isempty(xt), error('this mixture script is not for training'); end
  [yall,return_ind] = nagp_inputs(x,y,xt);
  [F,L,Qc,H,Pinf,Wnmf,D,N,lik_param] = nagp_stack_sources(w,x,ss,kernel1,kernel2,J);
  [model,A,Q] = nagp_model(F,L,Qc,H,Pinf,Wnmf,D,N,lik_param,true);
  tables = nagp_ihgp_tables(A,Q,H,true);
  o = nagp_opts(1,false,nagp_closure(mom,N),ep_fraction,ep_damping(1)*ones(1,ep_itts),ep_itts,'flags',8+1);
  [varargout{1:max(nargout,1)}] = nagp_call(nargout,0,return_ind,model,yall,o,tables);
end
