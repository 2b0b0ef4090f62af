"""cocons_amd -- MI355X-native dense hot path of the R package blasif/cocons.

Host-side mirror of the reference's operator interface for this path (same
names and argument meaning as the R functions) over the C ABI in
include/cocons_hip.h.  See DESIGN.md / INTEGRATION.md.
"""
from .host import (  # noqa: F401
    ASPECTS,
    CoconsFit,
    CoconsTaperFit,
    GetNeg2loglikelihood,
    GetNeg2loglikelihoodTaper,
    GetNeg2loglikelihoodTaperProfile,
    GetNeg2loglikelihoodProfile,
    GetNeg2loglikelihoodREML,
    GetNeg2loglikelihood_batch,
    cocoPredict_dense,
    cocoPredict_sparse,
    cocoSim_cond_dense,
    cocoSim_dense,
    cov_rns,
    cov_rns_classic,
    cov_rns_pred,
    cov_rns_taper,
    cov_rns_taper_pred,
    getBetas_profile,
    getHessian_dense,
    getModelLists,
    getPen,
    getScale,
    sumsmoothlone,
)
from ._lib import CholeskyError, CoconsHipError  # noqa: F401
