function [Esig,Vsig,Eft_mod,Varft_mod] = nagp_reconstruct(Eft,Varft,Wnmf,link,n_samples,seed)
% NAGP_RECONSTRUCT - what the drivers do with Eft, Varft (demo_toy_modulators_nmf.m:119-158): the reconstructed signal
% sig = sum_d (Wnmf*link(g))_d z_d and the modulator amplitudes link(g_n) under the independent posterior marginals
%
%   [Esig,Vsig,Eft_mod,Varft_mod] = nagp_reconstruct(Eft,Varft,Wnmf,link[,n_samples[,seed]])
%
% link: the driver's link handle (@(g)log(1+exp(g-shift)) or @(g)exp(g)); n_samples = 0 (default): the population values of
% the reference's sample statistics (Gauss-Hermite per modulator, closed-form combination); n_samples >= 2: the reference's
% estimator (250 draws there) on reproducible draws.
  if nargin < 5 || isempty(n_samples), n_samples = 0; end
  if nargin < 6 || isempty(seed), seed = 0; end
  [link_kind,link_shift] = nagp_link(link);
  [gx,gw] = gauher(32);                          % nodes / weights for the standard normal weight (reference file gauher.m)
  [Esig,Vsig,Eft_mod,Varft_mod] = nagp_mex('reconstruct',Eft,Varft,Wnmf,link_kind,link_shift,gx(:)',gw(:)',n_samples,seed);
end
