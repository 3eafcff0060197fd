function [M,P,K,MU,S,LH] = ekf_update1(M,P,y,H,R,h,V,param)
% EKF_UPDATE1 - first-order EKF measurement update of the reference ON THE GPU
%
% Same call as the reference's matlab/ekf_update1.m:48 (:106-109: one linearisation = iekf_update1 with iters = 1)
% for the dhandle / handle closures of gf_giekf_modulator_nmf*.m; see iekf_update1.m in this directory.

  if nargin < 5, error('Too few arguments'); end
  if nargin < 7, V = []; end
  if nargin < 8, param = []; end
  if nargout > 5
    [M,P,K,MU,S,LH] = iekf_update1(M,P,y,H,R,h,V,param,1);
  else
    [M,P,K,MU,S] = iekf_update1(M,P,y,H,R,h,V,param,1);
  end
end
