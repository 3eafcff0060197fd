function [link_kind,link_shift] = nagp_link(link)
% NAGP_LINK - enum + shift of a link function handle of the reference drivers
%   @(g)log(1+exp(g)), @(g)log(1+exp(g-1)), @(g)log(1+exp(g-mod_sparsity))  -> link_kind 0, link_shift
%   @(g)exp(g)                                                               -> link_kind 1
% (demo_toy_modulators.m:16-17, train_model.m:38, demo_toy_modulators_nmf_constraints.m:12); anything else is an error.
  s = regexprep(func2str(link),'\s','');
  link_kind = 0; link_shift = 0;
  tok = regexp(s,'^@\((\w+)\)log\(1\+exp\(\1(?:-([\w\.]+))?\)\)$','tokens','once');
  if ~isempty(tok)
    if numel(tok) > 1 && ~isempty(tok{2})
      v = str2double(tok{2});
      if isnan(v)                              % a captured variable, e.g. mod_sparsity
        lw = functions(link); lw = lw.workspace{1};
        if ~isfield(lw,tok{2}), error('nagp:closure','cannot resolve %s in the link function',tok{2}); end
        v = lw.(tok{2});
      end
      link_shift = double(v);
    end
  elseif ~isempty(regexp(s,'^@\((\w+)\)exp\(\1\)$','once'))
    link_kind = 1;
  else
    error('nagp:closure','link %s has no GPU implementation',s);
  end
end
