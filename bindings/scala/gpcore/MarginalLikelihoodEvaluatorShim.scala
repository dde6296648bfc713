package gp.classification

import breeze.linalg.{DenseMatrix, DenseVector}
import gp.classification.EpParameterEstimator.{AvgBasedStopCriterion, SiteParams}
import gpcore.Native
import utils.KernelRequisites._
import utils.MatrixUtils

/** Drop-in body for gp.classification.MarginalLikelihoodEvaluator (gp/classification/MarginalLikelihoodEvaluator.scala:13-76)
  * plus a correct batched mesh evaluator (MeshHyperParamsLogLikelihoodEvaluator.scala:26-40, whose result map is mis-keyed in
  * the reference -- SURVEY.md A23). */
class MarginalLikelihoodEvaluator(stopCriterion: EpParameterEstimator.stopCriterionFunc, kernelFunc: KernelFunc) {
  import MarginalLikelihoodEvaluator._
  import Native.{defaultCtx => ctx, dense, rethrowNotPd}

  /** what the batched device paths (HyperParamsOptimization / MeshHyperParamsLogLikelihoodEvaluator shims) need to know: the kernel,
    * and the eps of the stop criterion when it is the reference's AvgBasedStopCriterion (the only criterion the library evaluates
    * itself, per problem and sweep, operator precedence as written) */
  def kernel: KernelFunc = kernelFunc
  def stopEps: Option[Double] = stopCriterion match { case a: AvgBasedStopCriterion => Some(a.eps); case _ => None }

  def logLikelihoodWithKernelMatrixPassed(kernelMatrix: DenseMatrix[Double], targets: DenseVector[Int]): Double =
    new EpParameterEstimator(kernelMatrix, targets, stopCriterion).estimateSiteParams._1.marginalLogLikelihood.get

  def logLikelihoodWithoutGrad(trainInput: DenseMatrix[Double], targets: DenseVector[Int], hyperParams: DenseVector[Double]): Double =
    logLikelihoodWithKernelMatrixPassed(MatrixUtils.buildKernelMatrix(kernelFunc.changeHyperParams(hyperParams), trainInput), targets)

  // :33-44 -- EP on the device, then the gradient from the SAME device state (no second factorisation)
  def logLikelihood(trainInput: DenseMatrix[Double], targets: DenseVector[Int],
                    hyperParams: DenseVector[Double]): (Double, DenseVector[Double]) = {
    val newKernelFunc = kernelFunc.changeHyperParams(hyperParams)
    require(newKernelFunc.isInstanceOf[GaussianRbfKernel], "device EP gradient is implemented for GaussianRbfKernel")
    val x = dense(trainInput); val n = x.rows
    val k = MatrixUtils.buildKernelMatrix(newKernelFunc, trainInput)
    val ep = Native.epCreate(ctx, k.data, k.offset, n, k.majorStride, targets.toArray)
    try {
      var tau = new Array[Double](n); var nu = new Array[Double](n)
      var current = SiteParams(DenseVector(tau), DenseVector(nu)); var old = current
      var j = 0
      while (j == 0 || !stopCriterion(EpParameterEstimator.EpEstimationContext(currentParams = current, oldParams = old))) {
        old = current
        tau = new Array[Double](n); nu = new Array[Double](n)
        rethrowNotPd { Native.epSweep(ctx, ep, 1, n, tau, nu) }
        current = SiteParams(DenseVector(tau), DenseVector(nu))
        j += 1
      }
      val grad = new Array[Double](newKernelFunc.hyperParametersNum)
      Native.epLmlGradRbf(ctx, ep, x.data, x.offset, n, x.cols, x.majorStride, hyperParams.toArray, Native.strict, grad)   // :46-66
      (Native.epLml(ctx, ep, Native.strict), DenseVector(grad))
    } finally Native.epDestroy(ep)
  }

  // :46-66 with a caller-held (siteParams, L, K): reload the site parameters, one refactorisation, gradient
  def logLikelihoodDerivativesAfterHyperParams(optimInput: HyperParameterOptimInput, kernelFun: KernelFunc): DenseVector[Double] = {
    require(kernelFun.isInstanceOf[GaussianRbfKernel], "device EP gradient is implemented for GaussianRbfKernel")
    val k = dense(optimInput.kernelMatrix); val x = dense(optimInput.trainInput); val n = k.rows
    // the labels do not enter the gradient (:46-66 reads site parameters, L, K and X only)
    val ep = Native.epCreate(ctx, k.data, k.offset, n, k.majorStride, Array.fill(n)(1))
    try {
      rethrowNotPd {
        Native.epSetSiteParams(ctx, ep, n, optimInput.siteParams.tauSiteParams.toArray, optimInput.siteParams.niSiteParams.toArray)
      }
      val grad = new Array[Double](kernelFun.hyperParametersNum)
      Native.epLmlGradRbf(ctx, ep, x.data, x.offset, n, x.cols, x.majorStride, kernelFun.hyperParams.toDenseVector.toArray,
        Native.strict, grad)
      DenseVector(grad)
    } finally Native.epDestroy(ep)
  }

  /** EP log marginal likelihood at every row of `thetas` (B x (d+2), one hyper-parameter vector per row), by row index: the
    * leaves of MeshHyperParamsLogLikelihoodEvaluator.recEvaluate's grid in one library call (settings run concurrently on the
    * device).  Only for AvgBasedStopCriterion(eps) (the criterion spring-context.xml:53-55 wires); `maxSweeps` bounds a run. */
  def logLikelihoodOverMesh(trainInput: DenseMatrix[Double], targets: DenseVector[Int], thetas: DenseMatrix[Double],
                            eps: Double, maxSweeps: Int = 1000): DenseVector[Double] = {
    val x = dense(trainInput); val b = thetas.rows
    val flat = thetas.t.copy.data                                  // row-major B x P
    val lml = new Array[Double](b); val sweeps = new Array[Int](b); val info = new Array[Int](b)
    Native.epLmlRbfBatched(ctx, x.data, x.offset, x.rows, x.cols, x.majorStride, targets.toArray, flat, b, eps, maxSweeps,
      Native.strict, lml, sweeps, info)
    DenseVector(lml)                                               // NaN where I + S^1/2 K S^1/2 was not positive definite
  }
}

object MarginalLikelihoodEvaluator {
  type kernelAfterParamDerivative = (Double, DenseVector[Double], DenseVector[Double]) => Double
  type logLikelihoodAfterParamDerivative = (Double) => Double
  case class HyperParameterOptimInput(siteParams: SiteParams, lowerTriangular: DenseMatrix[Double],
                                      kernelMatrix: DenseMatrix[Double], trainInput: DenseMatrix[Double])
}
