function [varargout] = gf_giekf_modulator_nmf_constraints(w,x,y,ss,mom,xt,kernel1,kernel2,num_lik_params,D,N,g_iter,l_iter,...
                                                           constraints,w_fixed,tune_hypers,GradObj)
% GF_GIEKF_MODULATOR_NMF_CONSTRAINTS - gf_giekf_modulator_nmf with box-constrained, partially fixed hyper-parameters
%
% Same call as matlab/gf_giekf_modulator_nmf_constraints.m:1-2.  P is reset to Pinf at the start of every global
% iteration (:163-167; flag 2 = NAGP_FLAG_EKF_RESET_P).  xt empty: edata of ONE plain EKF pass with
% Q = Pinf - A*Pinf*A' (:332-480, GradObj 'off' as train_GTFNMF.m:198-201 calls it), eg = zeros.

  if nargin < 6, xt = []; end
  if nargin < 17, GradObj = 'off'; end
  [yall,return_ind] = nagp_inputs(x,y,xt);
  [lik_param,param1,param2,Wnmf] = nagp_unpack_constraints(w,w_fixed,tune_hypers,constraints,num_lik_params,D,N);
  [F,L,Qc,H,Pinf] = ss(x,param1,param2,kernel1,kernel2);
  [F,L,H,Pinf] = nagp_balance(F,L,H,Pinf);

  nlml = isempty(xt);
  if nlml && ~strcmpi(GradObj,'off'), error('nagp:giekf','only GradObj = ''off'' is served on the GPU'); end
  model = nagp_model(F,L,Qc,H,Pinf,Wnmf,D,N,lik_param,false,nlml);
  if nlml, g_iter = 1; l_iter = 1; end
  o = nagp_opts(2,nlml,[],0.5,[],g_iter,'l_iter',l_iter,'flags',2);
  [varargout{1:max(nargout,1)}] = nagp_call(nargout,numel(w),return_ind,model,yall,o);
end
