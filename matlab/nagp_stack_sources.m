function [F,L,Qc,H,Pinf,Wnmf,D,N,lik_param] = nagp_stack_sources(w,x,ss,kernel1,kernel2,J)
% NAGP_STACK_SOURCES - the model set-up shared by experiments/gf_ep_mods_nmf_mixture.m:89-128 and
% experiments/ihgp_ep_mods_nmf_mixture.m:86-125: J GT-NMF models stacked block-diagonally, the sub-band blocks of all
% sources first, then all modulator blocks; Wnmf = blkdiag(W_1, ..., W_J).
%   w = {lik_param, {param1_j}, {param2_j}, {W_j}}

  lik_param = w{1};
  F_z=[];L_z=[];Qc_z=[];H_z=[];Pinf_z=[]; F_g=[];L_g=[];Qc_g=[];H_g=[];Pinf_g=[];
  Wnmf = []; D = 0; N = 0;
  for j = 1:J
    param1 = w{2}{j}; D_ = length(param1)/3; D = D + D_;
    param2 = w{3}{j}; N_ = length(param2)/2; N = N + N_;
    Wnmf = blkdiag(Wnmf,w{4}{j});
    cf_to_ss1 = str2func(strcat('cf_',kernel1{j},'_to_ss'));
    tau1 = size(cf_to_ss1(1,1,6),1); tau2 = 2; z_tau = tau1*tau2;
    [F_j,L_j,Qc_j,H_j,Pinf_j] = ss(x,param1,param2,kernel1{j},kernel2{j});
    iz = 1:D_*z_tau; ig = D_*z_tau+1:size(F_j,1);
    F_z = blkdiag(F_z,F_j(iz,iz));       F_g = blkdiag(F_g,F_j(ig,ig));
    L_z = blkdiag(L_z,L_j(iz,1:D_*tau2)); L_g = blkdiag(L_g,L_j(ig,D_*tau2+1:end));
    Qc_z = blkdiag(Qc_z,Qc_j(1:D_*tau2,1:D_*tau2)); Qc_g = blkdiag(Qc_g,Qc_j(D_*tau2+1:end,D_*tau2+1:end));
    H_z = blkdiag(H_z,H_j(1:D_,iz));     H_g = blkdiag(H_g,H_j(D_+1:end,ig));
    Pinf_z = blkdiag(Pinf_z,Pinf_j(iz,iz)); Pinf_g = blkdiag(Pinf_g,Pinf_j(ig,ig));
  end
  F = blkdiag(F_z,F_g); L = blkdiag(L_z,L_g); Qc = blkdiag(Qc_z,Qc_g); H = blkdiag(H_z,H_g); Pinf = blkdiag(Pinf_z,Pinf_g);
end
