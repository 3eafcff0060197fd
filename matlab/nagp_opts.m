function o = nagp_opts(kind, nlml, c, ep_fraction, ep_damping, ep_itts, varargin)
% NAGP_OPTS - options struct of nagp_mex (include/nagp.h: nagp_opts)
%   kind 0 gf_ep_*, 1 ihgp_*, 2 gf_giekf_*;  nlml: true when xt is empty (the fminunc objective);
%   c: nagp_closure(...) or [] (EKF);  further name/value pairs: 'l_iter', 'predict_at_k1', 'flags', 'device'
  o = struct('kind',kind,'mode',double(nlml),'lik_kind',1,'link_kind',0,'link_shift',0,'wn',[],'xn_unscaled',[], ...
             'ep_fraction',ep_fraction,'ep_damping',[],'ep_itts',ep_itts,'l_iter',0,'predict_at_k1',0,'flags',0,'device',0);
  if ~isempty(c)
    o.lik_kind = c.lik_kind; o.link_kind = c.link_kind; o.link_shift = c.link_shift;
    o.wn = c.wn; o.xn_unscaled = c.xn_unscaled;
    o.ep_damping = nagp_damping(ep_damping,ep_itts);
  end
  for i = 1:2:numel(varargin), o.(varargin{i}) = varargin{i+1}; end
end
