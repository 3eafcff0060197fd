function [varargout] = ihgp_ep_modulator_nmf(w,x,y,ss,mom,xt,kernel1,kernel2,num_lik_params,D,N,ep_fraction,ep_damping,ep_itts)
% IHGP_EP_MODULATOR_NMF - infinite-horizon (steady-state) GT-NMF inference on the GPU
%
% Same call as matlab/ihgp_ep_modulator_nmf.m:1.  Set-up as there: log parameters (:72-75), balance ON (:81-87),
% lti_disc and Q = (Q+Q')/2 (:96-97), the DARE look-up tables (:99-191, nagp_ihgp_tables).  The sweep loop
% (:223-454) runs in libnagp.so.  Only prediction exists: the reference's objective branch cannot run (SURVEY C-11).

  if nargin < 12, ep_fraction = 0.5; end
  if nargin < 14, ep_itts = 30; end
  if isempty(xt), error('nagp:ihgp','the infinite-horizon functions only predict (the nlml branch of the reference is broken)'); end
  [yall,return_ind] = nagp_inputs(x,y,xt);
  n0 = num_lik_params;
  lik_param = w(1:n0);
  param1 = exp(w(n0+1:n0+3*D));
  param2 = exp(w(n0+3*D+1:n0+3*D+2*N));
  Wnmf = reshape(exp(w(n0+3*D+2*N+1:end)),[D,N]);
  [F,L,Qc,H,Pinf] = ss(x,param1,param2,kernel1,kernel2);
  [F,L,H,Pinf] = nagp_balance(F,L,H,Pinf);

  [model,A,Q] = nagp_model(F,L,Qc,H,Pinf,Wnmf,D,N,lik_param,true);
  tables = nagp_ihgp_tables(A,Q,H,true);
  o = nagp_opts(1,false,nagp_closure(mom,N),ep_fraction,ep_damping,ep_itts);
  [varargout{1:max(nargout,1)}] = nagp_call(nargout,numel(w),return_ind,model,yall,o,tables);
end
