"""nagp -- host-side mirror of the reference's hot-path interface over libnagp.so (HIP, gfx950).

    from nagp import gf_ep_modulator_nmf, Mom, SSHandle
"""
from ._lib import build, lib, NagpError, LIB_PATH  # noqa: F401
from .api import (Mom, SSHandle, gf_ep_modulator, gf_ep_modulator_nmf, gf_ep_modulator_nmf_constraints,  # noqa: F401
                  ihgp_ep_modulator_nmf, ihgp_ep_modulator_nmf_constraints, gf_giekf_modulator_nmf,
                  gf_giekf_modulator_nmf_constraints, MeasModel, ekf_update1, iekf_update1,
                  gf_ep_mods_nmf_mixture, ihgp_ep_mods_nmf_mixture)
from .ss import ss_modulators, ss_modulators_nmf, lti_disc, sigmoid, inv_sigmoid  # noqa: F401
from .cubature import utp_ws, gauher, mvhermgauss_unit  # noqa: F401
from .plan import Plan, batch_run, batch_partition  # noqa: F401
from .fastfb import get_disc_model, kernel_ss_kalmanFastFB  # noqa: F401
from .train import nlml_batch, fd_value_and_gradient  # noqa: F401
from .recon import reconstruct_signal  # noqa: F401
