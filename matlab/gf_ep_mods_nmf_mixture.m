function [varargout] = gf_ep_mods_nmf_mixture(w,x,y,ss,mom,xt,kernel1,kernel2,J,ep_fraction,ep_damping,ep_itts)
% GF_EP_MODS_NMF_MIXTURE - source separation: J stacked GT-NMF models by Power EP (full covariance) ON THE GPU
%
% Same call as the reference's matlab/experiments/gf_ep_mods_nmf_mixture.m:1.  The loops :130-349 run in libnagp.so
% under the older EP rule of this file (flag 8 = NAGP_FLAG_MIXTURE_RULE: `mom` at power ep_fraction in the filter's ADF
% step too, site <- (1-d) site + d/ep_fraction (...), clamp in the filter pass only, :183-195, 277-284).  The `mom`
% these files call takes six arguments (the power is baked into the closure, source_sep_piano.m:93); the closure's
% captured ep_fraction, when there is one, must equal the ep_fraction argument.  Prediction only (:376).

  if nargin < 10, ep_fraction = 0.5; end
  if nargin < 11, ep_damping = 0.1; end
  if nargin < 12, ep_itts = 30; end
  if isempty(xt), error('this mixture script is not for training'); end
  [yall,return_ind] = nagp_inputs(x,y,xt);
  [F,L,Qc,H,Pinf,Wnmf,D,N,lik_param] = nagp_stack_sources(w,x,ss,kernel1,kernel2,J);
  model = nagp_model(F,L,Qc,H,Pinf,Wnmf,D,N,lik_param);
  o = nagp_opts(0,false,nagp_closure(mom,N),ep_fraction,ep_damping(1)*ones(1,ep_itts),ep_itts,'flags',8);
  [varargout{1:max(nargout,1)}] = nagp_call(nargout,0,return_ind,model,yall,o);
end
