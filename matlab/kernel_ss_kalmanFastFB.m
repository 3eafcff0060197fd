function [lik,Xfin,Pfin,varargout] = kernel_ss_kalmanFastFB(A,Q,C,P0,K,vary,y,varargin)
% KERNEL_SS_KALMANFASTFB - infinite-horizon Kalman filter / steady-state RTS smoother of the stationary filterbank ON THE GPU
%
% Same call as the reference's matlab/unifying_prob_tf/kernel_ss_kalmanFastFB.m:1 (the step before the hot path in every
% real-audio script, e.g. train_GTFNMF.m:56-65).  The set-up lines of the .m stay here (dare, stationary gain :46-59,
% smoother gain and covariance :127-132); the two O(T) loops :83-110 and :134-151 run in libnagp.so (nagp_fastfb_run:
% parallel in time over spans).  Pfin holds the steady-state covariance in every slice (the filter's in the last one),
% as the reference stores it.

  T = length(y);
  if nargin <= 7, verbose = 0; else, verbose = varargin{1}; end                %#ok<NASGU>
  if nargin <= 8, KF = 0; else, KF = varargin{2}; end
  H = C; R = vary;
  try
    PP = dare(A',H',Q,R);
    S = H*PP*H' + R;
  catch
    error('Unstable DARE solution!')
  end
  Kg = PP*H'/S;
  AKHA = A - Kg*H*A;
  PF2 = PP - Kg*H*PP;
  HA = H*A;
  G = []; Psm = [];
  if KF ~= 1
    G = PF2*A'/PP;
    QQ = PF2 - G*PP*G'; QQ = (QQ+QQ')/2;
    Psm = dare(G',zeros(size(QQ)),QQ);
  end
  [MS,sum_v2] = nagp_mex('fastfb', A, AKHA, HA(:), Kg(:), G, y(:));
  lik = -(0.5*log(2*pi)*T + 0.5*log(S)*T + 0.5*sum_v2/S);
  Xfin = reshape(MS,[1 size(MS)]);
  if isempty(Psm), Pfin = repmat(PF2,[1 1 T]); else, Pfin = repmat(Psm,[1 1 T]); Pfin(:,:,T) = PF2; end
end
