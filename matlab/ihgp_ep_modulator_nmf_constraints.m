function [varargout] = ihgp_ep_modulator_nmf_constraints(w,x,y,ss,mom,xt,kernel1,kernel2,num_lik_params,D,N,ep_fraction,ep_damping,ep_itts,...
                                                          constraints,w_fixed,tune_hypers)
% IHGP_EP_MODULATOR_NMF_CONSTRAINTS - ihgp_ep_modulator_nmf with box-constrained, partially fixed hyper-parameters
%
% Same call as matlab/ihgp_ep_modulator_nmf_constraints.m:1-2.  Differences to the plain variant reproduced by
% flag 1 (NAGP_FLAG_IHGP_CONSTRAINTS): R starts at zero (:243) and Varft is returned without abs() (:517-518).

  if isempty(xt), error('nagp:ihgp','the infinite-horizon functions only predict (the nlml branch of the reference is broken)'); end
  [yall,return_ind] = nagp_inputs(x,y,xt);
  [lik_param,param1,param2,Wnmf] = nagp_unpack_constraints(w,w_fixed,tune_hypers,constraints,num_lik_params,D,N);
  [F,L,Qc,H,Pinf] = ss(x,param1,param2,kernel1,kernel2);
  [F,L,H,Pinf] = nagp_balance(F,L,H,Pinf);

  [model,A,Q] = nagp_model(F,L,Qc,H,Pinf,Wnmf,D,N,lik_param,true);
  tables = nagp_ihgp_tables(A,Q,H,true);
  o = nagp_opts(1,false,nagp_closure(mom,N),ep_fraction,ep_damping,ep_itts,'flags',1);
  [varargout{1:max(nargout,1)}] = nagp_call(nargout,numel(w),return_ind,model,yall,o,tables);
end
