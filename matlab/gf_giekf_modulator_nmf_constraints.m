function [varargout] = gf_giekf_modulator_nmf_constraints(w,x,y,ss,mom,xt,kernel1,kernel2,num_lik_params,D,N,g_iter,l_iter,...
                                                           constraints,w_fixed,tune_hypers,GradObj)
% GF_GIEKF_MODULATOR_NMF_CONSTRAINTS - gf_giekf_modulator_nmf with box-constrained, partially fixed hyper-parameters
%
% Same call as matlab/gf_giekf_modulator_nmf_constraints.m:1-2.  P is reset to Pinf at the start of every global
% iteration (:163-167; flag 2 = NAGP_FLAG_EKF_RESET_P).  xt empty: edata of ONE plain EKF pass with
% Q = Pinf - A*Pinf*A' (:332-480); GradObj 'off' (as train_GTFNMF.m:198-201 calls it): eg = zeros; 'on': the gradient
% recursion of the same lines on the GPU (nagp_giekf_nlml_grad), as written in the reference.

  if nargin < 6, xt = []; end
  if nargin < 17, GradObj = 'off'; end
  [yall,return_ind] = nagp_inputs(x,y,xt);
  [lik_param,param1,param2,Wnmf] = nagp_unpack_constraints(w,w_fixed,tune_hypers,constraints,num_lik_params,D,N);
  nlml = isempty(xt);
  if nlml && strcmpi(GradObj,'on')
    % :332-480 as written: the derivative stacks of ss_modulators_nmf (unbalanced, :117-119 commented out) with a zero slice for
    % the noise parameter in front (:121-125), AA_j = expm([F 0; dF_j F]) (:355-366), dQ_j (:392-394)
    [F,L,Qc,H,Pinf,dF,~,dPinf] = ss(x,param1,param2,kernel1,kernel2);
    [F,L,H,Pinf] = nagp_balance(F,L,H,Pinf);
    d = size(F,1);
    dF = cat(3,zeros(d),dF); dPinf = cat(3,zeros(d),dPinf); np_ = size(dF,3);
    if numel(w) < np_, error('MATLAB:badsubscript','Index exceeds the number of array elements (%d).',numel(w)); end   % gdata(j), :453-457
    model = nagp_model(F,L,Qc,H,Pinf,Wnmf,D,N,lik_param,false,true);
    A = model.A; dA = zeros(d,d,np_); dQ = zeros(d,d,np_);
    for j = 1:np_
      AA = expm([F zeros(d); dF(:,:,j) F]);
      dA(:,:,j) = AA(d+1:end,1:d);
      X = dA(:,:,j)*Pinf*A';
      dQ(:,:,j) = dPinf(:,:,j) - X - A*dPinf(:,:,j)*A' - X';
    end
    dR = zeros(1,np_); dR(1) = 1;
    jj = (1:np_) - (np_ - D*N);                                 % > 0: the slices that take funhd(.,W_) (:440-444)
    [e,g] = nagp_mex('giekf_grad',model,yall,dA,dQ,dPinf,dR,int32(jj <= 0),int32(max(jj,0) - 1),int32(zeros(1,np_)));
    eg = zeros(1,numel(w)); eg(1:np_) = g;
    varargout = {e,eg};
    return
  end
  [F,L,Qc,H,Pinf] = ss(x,param1,param2,kernel1,kernel2);
  [F,L,H,Pinf] = nagp_balance(F,L,H,Pinf);

  if nlml && ~strcmpi(GradObj,'off'), error('nagp:giekf','GradObj must be ''off'' or ''on'''); end
  model = nagp_model(F,L,Qc,H,Pinf,Wnmf,D,N,lik_param,false,nlml);
  if nlml, g_iter = 1; l_iter = 1; end
  o = nagp_opts(2,nlml,[],0.5,[],g_iter,'l_iter',l_iter,'flags',2);
  [varargout{1:max(nargout,1)}] = nagp_call(nargout,numel(w),return_ind,model,yall,o);
end
