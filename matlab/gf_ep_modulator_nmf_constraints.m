function [varargout] = gf_ep_modulator_nmf_constraints(w,x,y,ss,mom,xt,kernel1,kernel2,num_lik_params,D,N,ep_fraction,ep_damping,ep_itts,...
                                                        constraints,w_fixed,tune_hypers)
% GF_EP_MODULATOR_NMF_CONSTRAINTS - gf_ep_modulator_nmf with box-constrained, partially fixed hyper-parameters
%
% Same call as matlab/gf_ep_modulator_nmf_constraints.m:1-2; sigmoid-constrained unpacking (:75-110), balance ON
% (:115-121); the EP loop (:148-334 / :436-573) runs in libnagp.so.

  if nargin < 6, xt = []; end
  [yall,return_ind] = nagp_inputs(x,y,xt);
  [lik_param,param1,param2,Wnmf] = nagp_unpack_constraints(w,w_fixed,tune_hypers,constraints,num_lik_params,D,N);
  [F,L,Qc,H,Pinf] = ss(x,param1,param2,kernel1,kernel2);
  [F,L,H,Pinf] = nagp_balance(F,L,H,Pinf);

  model = nagp_model(F,L,Qc,H,Pinf,Wnmf,D,N,lik_param);
  o = nagp_opts(0,isempty(xt),nagp_closure(mom,N),ep_fraction,ep_damping,ep_itts);
  [varargout{1:max(nargout,1)}] = nagp_call(nargout,numel(w),return_ind,model,yall,o);
end
