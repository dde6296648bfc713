"""gp_algos_amd -- MI355X-native Gaussian-process hot path behind the reference's API names.

Layout mirrors the reference packages (astroHaoPeng/gp_algos, src/main/scala):
    utils.kernel_requisites   <- utils/KernelRequisites.scala   (GaussianRbfParams, GaussianRbfKernel, KernelFunc)
    utils.matrix_utils        <- utils/MatrixUtils.scala        (buildKernelMatrix, forwardSolve, backSolve, invTriangular)
    utils.stats_utils         <- utils/StatsUtils.scala         (pnorm, dnorm, GaussianDistribution)
    gp.regression.gp_predictor        <- gp/regression/GpPredictor.scala
    gp.classification.*               <- gp/classification/{EpParameterEstimator,GpClassifier,MarginalLikelihoodEvaluator,...}.scala
    optimization.optimization         <- optimization/Optimization.scala
All numerics run in libgpcore.so (hand-written HIP for gfx950) through the C-ABI of include/gpcore.h;
there is no CPU fallback.
"""
_default_ctx = None


def default_context(device=0):
    """Process-wide device context used by the reference-API mirror (created on first use)."""
    global _default_ctx
    if _default_ctx is None or getattr(_default_ctx, "h", None) is None:
        from .core import Context
        _default_ctx = Context(device)
    return _default_ctx


def set_default_context(ctx):
    global _default_ctx
    _default_ctx = ctx
