function [varargout] = gf_giekf_modulator_nmf(w,x,y,ss,mom,xt,kernel1,kernel2,num_lik_params,D,N,g_iter,l_iter,GradObj)
% GF_GIEKF_MODULATOR_NMF - globally iterated extended Kalman filter / RTS smoother on the GPU
%
% Same call as matlab/gf_giekf_modulator_nmf.m:1-2 (mom is accepted and unused, :13).  Log parameters, balance ON
% (:78-84); g_iter passes of the EKF filter (iekf_update1 with l_iter inner iterations) and the RTS smoother
% (:126-221) run in libnagp.so.  The measurement Jacobian is the corrected one of the constraints variant
% (SURVEY C-13).  xt empty: the energy of one EKF pass, GradObj 'off' only (see DESIGN.md for the gradient branch).

  if nargin < 6, xt = []; end
  if nargin < 12 || isempty(g_iter), g_iter = 1; end
  if nargin < 13 || isempty(l_iter), l_iter = 1; end
  if nargin < 14, GradObj = 'off'; end
  [yall,return_ind] = nagp_inputs(x,y,xt);
  n0 = num_lik_params;
  lik_param = w(1:n0);
  param1 = exp(w(n0+1:n0+3*D));
  param2 = exp(w(n0+3*D+1:n0+3*D+2*N));
  Wnmf = reshape(exp(w(n0+3*D+2*N+1:end)),[D,N]);
  [F,L,Qc,H,Pinf] = ss(x,param1,param2,kernel1,kernel2);
  [F,L,H,Pinf] = nagp_balance(F,L,H,Pinf);

  nlml = isempty(xt);
  if nlml && ~strcmpi(GradObj,'off'), error('nagp:giekf','only GradObj = ''off'' is served on the GPU'); end
  model = nagp_model(F,L,Qc,H,Pinf,Wnmf,D,N,lik_param,false,nlml);           % nlml: Q = Pinf - A*Pinf*A'
  if nlml, g_iter = 1; l_iter = 1; end
  o = nagp_opts(2,nlml,[],0.5,[],g_iter,'l_iter',l_iter);
  [varargout{1:max(nargout,1)}] = nagp_call(nargout,numel(w),return_ind,model,yall,o);
end
