package gp.classification

import breeze.linalg.{DenseMatrix, DenseVector}
import breeze.numerics.abs
import gpcore.Native

/** Drop-in body for gp.classification.EpParameterEstimator (gp/classification/EpParameterEstimator.scala:11-109,181-202).
  * The site loop, the per-sweep refactorisation and the EP log marginal likelihood run in libgpcore.so (gp_ep_*); the stop
  * criterion stays an arbitrary Scala function of (old, current) site parameters, evaluated on the host between sweeps
  * exactly where the reference evaluates it (:40: before every sweep but the first). */
class EpParameterEstimator(kernelMatrix: DenseMatrix[Double], targets: DenseVector[Int],
                           stopCriterion: EpParameterEstimator.stopCriterionFunc) {
  import EpParameterEstimator._
  import Native.{defaultCtx => ctx, dense, rethrowNotPd}

  require(kernelMatrix.rows == targets.length)

  def estimateSiteParams: (SiteParams, DenseMatrix[Double]) = {
    val n = kernelMatrix.rows
    val k = dense(kernelMatrix)
    val ep = Native.epCreate(ctx, k.data, k.offset, n, k.majorStride, targets.toArray)   // {-1,+1} checked by the library
    try {
      var tau = new Array[Double](n); var nu = new Array[Double](n)
      var current = SiteParams(tauSiteParams = DenseVector(tau), niSiteParams = DenseVector(nu))
      var old = current
      var j = 0
      while (j == 0 || !stopCriterion(EpEstimationContext(currentParams = current, oldParams = old))) {
        old = current
        tau = new Array[Double](n); nu = new Array[Double](n)
        rethrowNotPd { Native.epSweep(ctx, ep, 1, n, tau, nu) }          // cholesky(I + S^1/2 K S^1/2) throws on a negative site precision
        current = SiteParams(tauSiteParams = DenseVector(tau), niSiteParams = DenseVector(nu))
        j += 1
      }
      val lml = Native.epLml(ctx, ep, Native.strict)                      // :71-96 (as compiled unless -Dgpcore.strict=false)
      val l = new Array[Double](n * n)
      Native.epGet(ctx, ep, 0, l, n)
      (current.copy(marginalLogLikelihood = Some(lml)), new DenseMatrix(n, n, l))
    } finally Native.epDestroy(ep)
  }
}

object EpParameterEstimator {
  type stopCriterionFunc = EpEstimationContext => Boolean

  case class SiteParams(tauSiteParams: DenseVector[Double], niSiteParams: DenseVector[Double],
                        marginalLogLikelihood: Option[Double] = None)
  case class CavityDistributionParams(tauParams: DenseVector[Double], niParams: DenseVector[Double])
  case class EpEstimationContext(oldParams: SiteParams, currentParams: SiteParams)

  /** `val eps` (the reference keeps it as a plain constructor parameter, :187): the batched device paths need to read it back */
  class AvgBasedStopCriterion(val eps: Double) extends stopCriterionFunc {
    def apply(context: EpEstimationContext): Boolean =
      abs(avgBetweenSiteParams(context.oldParams, context.currentParams)) < eps
  }

  // :195-202, operator precedence as written: (sum / 2) * n
  def avgBetweenSiteParams(oldParams: SiteParams, currentParams: SiteParams): Double = {
    val sum = (0 until oldParams.niSiteParams.length).foldLeft(0.0) { case (acc, i) =>
      acc + (currentParams.niSiteParams(i) - oldParams.niSiteParams(i)) + (currentParams.tauSiteParams(i) - oldParams.tauSiteParams(i))
    }
    sum / 2 * currentParams.niSiteParams.length
  }
}
