function varargout = nagp_outputs(nout, return_ind, Eft, Varft, out)
% NAGP_OUTPUTS - the return convention of the prediction branch of the reference functions
% (gf_ep_modulator_nmf.m:313-350): {Eft,Varft} or {Eft,Varft,Covft,lb,ub,out}, test columns only.
  Eft = Eft(:,return_ind); Varft = Varft(:,return_ind);
  varargout = {Eft,Varft};
  if nout > 3
    lb = Eft - 1.96*sqrt(Varft);
    ub = Eft + 1.96*sqrt(Varft);
    varargout = {Eft,Varft,[],lb,ub,out};
  elseif nout == 3
    varargout = {Eft,Varft,[]};
  end
end
