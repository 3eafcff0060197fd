function [yall, return_ind] = nagp_inputs(x, y, xt)
% NAGP_INPUTS - merge training and test inputs as every reference function does (gf_ep_modulator_nmf.m:58-66):
% observations at the test-only inputs are NaN, inputs are made unique ('first') and ascending, return_ind maps
% the test inputs to columns of the outputs.  The time step is one sample (SURVEY C-19).
  xall = [x(:); xt(:)];
  yall = [y(:); nan(numel(xt),1)];
  [~,sort_ind,return_ind] = unique(xall,'first');
  yall = yall(sort_ind);
  return_ind = return_ind(end-numel(xt)+1:end);
end
