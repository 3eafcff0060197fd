function [varargout] = gf_ep_modulator_nmf(w,x,y,ss,mom,xt,kernel1,kernel2,num_lik_params,D,N,ep_fraction,ep_damping,ep_itts)
% GF_EP_MODULATOR_NMF - GT-NMF model by Power EP: Kalman filter + RTS smoother + site refresh ON THE GPU
%
% Same call as the reference's matlab/gf_ep_modulator_nmf.m:1 (put this directory before the reference's on the
% path; the reference's ss_modulators_nmf, lti_disc, utp_ws, ... are still used for the set-up).  The loop
% `for itt=1:ep_itts` (:113-283 when predicting, :384-522 for the objective) runs in libnagp.so through nagp_mex.
%
%   [Eft,Varft,Covft,lb,ub,out] = gf_ep_modulator_nmf(w,x,y,ss,mom,xt,...)     xt not empty
%   [e,eg]                      = gf_ep_modulator_nmf(w,x,y,ss,mom,[],...)     negative log marginal likelihood, eg = 0

  if nargin < 6, xt = []; end
  if nargin < 12, ep_fraction = 0.5; end
  if nargin < 14, ep_itts = 30; end
  [yall,return_ind] = nagp_inputs(x,y,xt);

  % log-transformed parameters (:72-75); balance stays off in this variant (:80)
  n0 = num_lik_params;
  lik_param = w(1:n0);
  param1 = exp(w(n0+1:n0+3*D));
  param2 = exp(w(n0+3*D+1:n0+3*D+2*N));
  Wnmf = reshape(exp(w(n0+3*D+2*N+1:end)),[D,N]);
  [F,L,Qc,H,Pinf] = ss(x,param1,param2,kernel1,kernel2);

  model = nagp_model(F,L,Qc,H,Pinf,Wnmf,D,N,lik_param);
  o = nagp_opts(0,isempty(xt),nagp_closure(mom,N),ep_fraction,ep_damping,ep_itts);
  [varargout{1:max(nargout,1)}] = nagp_call(nargout,numel(w),return_ind,model,yall,o);
end
