function [varargout] = gf_ep_modulator(w,x,y,ss,mom,xt,kernel1,kernel2,num_lik_params,ep_fraction,ep_damping,ep_itts)
% GF_EP_MODULATOR - one modulator per sub-band (no NMF mixing), Power EP on the GPU
%
% Same call as matlab/gf_ep_modulator.m:1.  w = [lik; log(var_fast, len_fast, omega, var_slow, len_slow)] with D
% entries per group; balance ON (:75-81); predict mode also predicts at the first step (:131-133, SURVEY C-8);
% the sites are the 2D rows of H (C-20) and the likelihood is likModulatorPower.

  if nargin < 6, xt = []; end
  if nargin < 10, ep_fraction = 0.5; end
  if nargin < 12, ep_itts = 30; end
  [yall,return_ind] = nagp_inputs(x,y,xt);
  lik_param = w(1:num_lik_params);
  param = exp(w(num_lik_params+1:end));
  D = numel(param)/5;
  [F,L,Qc,H,Pinf] = ss(x,param,kernel1,kernel2);
  [F,L,H,Pinf] = nagp_balance(F,L,H,Pinf);

  model = nagp_model(F,L,Qc,H,Pinf,[],D,D,lik_param);
  o = nagp_opts(0,isempty(xt),nagp_closure(mom,D),ep_fraction,ep_damping,ep_itts,'predict_at_k1',double(~isempty(xt)));
  [varargout{1:max(nargout,1)}] = nagp_call(nargout,numel(w),return_ind,model,yall,o);
end
