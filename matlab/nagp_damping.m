function d = nagp_damping(ep_damping, ep_itts)
% NAGP_DAMPING - one damping value per EP sweep; the experiment scripts pass a scalar with ep_itts > 1
% (noise_reduction_speech.m:29,94 -- the reference would index past it, SURVEY C-15): broadcast.
  d = ep_damping(:)';
  if numel(d) == 1, d = repmat(d,1,ep_itts); end
  if numel(d) < ep_itts, error('nagp:arg','ep_damping has fewer than ep_itts entries'); end
  d = d(1:ep_itts);
end
