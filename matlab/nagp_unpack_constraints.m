function [lik_param, param1, param2, Wnmf] = nagp_unpack_constraints(w, w_fixed, tune_hypers, constraints, num_lik_params, D, N)
% NAGP_UNPACK_CONSTRAINTS - box-constrained hyper-parameters, split into tuned (w) and fixed (w_fixed) groups
% (gf_ep_modulator_nmf_constraints.m:75-110).  tune_hypers(1..7) selects, per group
%   [likelihood, var_fast, len_fast, omega, var_slow, len_slow, W],
% whether it is read from w or from w_fixed; constraints(1..6,:) = [lo hi] of the sigmoid map of groups 2..7
% (sigmoid.m:17-19: y = (hi-lo)/(1+exp(-x)) + lo).
  src = {w_fixed(:), w(:)};               % index 1: fixed, 2: tuned
  pos = [0 0];
  cnt = [num_lik_params, D, D, D, N, N, D*N];
  vals = cell(1,7);
  for g = 1:7
    s = double(tune_hypers(g) ~= 0) + 1;
    if g == 7
      v = src{s}(pos(s)+1:end);
    else
      v = src{s}(pos(s)+1:pos(s)+cnt(g)); pos(s) = pos(s) + cnt(g);
    end
    if g > 1
      v = sigmoid(v,constraints(g-1,:));     % the reference's own helper (matlab/sigmoid.m)
    end
    vals{g} = v;
  end
  lik_param = vals{1};
  param1 = [vals{2}; vals{3}; vals{4}];
  param2 = [vals{5}; vals{6}];
  Wnmf = reshape(vals{7},[D,N]);
end
