function [nlZ_total,Eft,Varft,nlZ] = nagp_batch(models,ys,o,n_gpus,tables)
% NAGP_BATCH - independent problems of one shape (audio segments, or the numel(w)+1 objective evaluations of one
% fminunc iteration, train_GTFNMF.m:186-201) spread over the GPUs of the node in ONE call
%
%   [nlZ_total,Eft,Varft,nlZ] = nagp_batch(models,ys,o,n_gpus[,tables])
%
% models: cell array of nagp_model(...) structs; ys: cell array of observation vectors of one length; o: nagp_opts(...);
% tables: cell array of nagp_ihgp_tables(...) structs (infinite-horizon kind).  Problem i runs on GPU mod(i-1,n_gpus);
% nlZ_total(itt) = sum over ALL problems of -sum_k lZ_k (gf_ep_modulator_nmf.m:187, 277, 525), all-reduced over the GPUs
% with RCCL inside libnagp.so (nagp_batch_run).  Eft, Varft, nlZ are cell arrays, one entry per problem.
  if nargin > 4 && ~isempty(tables)
    [nlZ_total,Eft,Varft,nlZ] = nagp_mex('batch',models,ys,o,n_gpus,tables);
  else
    [nlZ_total,Eft,Varft,nlZ] = nagp_mex('batch',models,ys,o,n_gpus);
  end
end
