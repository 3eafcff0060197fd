function varargout = nagp_call(nout, nw, return_ind, model, yall, o, tables)
% NAGP_CALL - one trip through the MEX gateway and the return convention shared by all reference functions:
% prediction ({Eft,Varft[,Covft,lb,ub,out]}, gf_ep_modulator_nmf.m:313-350) when o.mode == 0, otherwise
% {e, eg} with e = -sum(lZ) and an all-zero gradient (:363, :531).
  args = {model,yall,o};
  if nargin > 6 && ~isempty(tables), args{end+1} = tables; end
  if o.mode ~= 0
    [~,~,~,~,~,~,nlZ] = nagp_mex(args{:});
    varargout = {nlZ(1), zeros(1,nw)};
    return
  end
  if nout > 5
    if o.kind == 1                                            % no covariances on the infinite-horizon path
      [Eft,Varft,ttau,tnu,R,lZ,nlZ,mdM,mdP,cnt,MS] = nagp_mex(args{:}); PS = [];
    else
      [Eft,Varft,ttau,tnu,R,lZ,nlZ,mdM,mdP,cnt,MS,PS] = nagp_mex(args{:});
    end
    out = struct('tnu',tnu,'ttau',ttau,'lZ',lZ,'R',R,'nlZ',nlZ,'maxDiffM',mdM,'maxDiffP',mdP,'MS',MS,'PS',PS, ...
                 'counters',struct('chol_retries',cnt(1),'clamped',cnt(2),'nan_obs',cnt(3),'not_pd',cnt(4)));
  else
    [Eft,Varft] = nagp_mex(args{:}); out = [];
  end
  [varargout{1:max(nout,2)}] = nagp_outputs(max(nout,2), return_ind, Eft, Varft, out);
  varargout = varargout(1:max(nout,1));
end
