package dynamicalsystems.filtering

import breeze.linalg.{diag, DenseMatrix, DenseVector}
import dynamicalsystems.filtering.SsmTypeDefinitions.SeriesGenerationData
import gp.optimization.GPOptimizer
import gp.regression.GpPredictor
import gpcore.Native
import org.slf4j.LoggerFactory
import utils.KernelRequisites.{GaussianRbfKernel, KernelFunc, KernelFuncHyperParams}
import utils.StatsUtils._

/** Drop-in body for dynamicalsystems.filtering.GPUnscentedKalmanFilter (GPUnscentedKalmanFilter.scala:15-152).  The public
  * methods and the unscented recursion it inherits are the reference's; what moves to the device is the GP state-space model that
  * `learnNewSsmModelWithNoises` (:63-96) builds -- SURVEY.md 8(f) rank 3.
  *
  * The reference learns one GP per hidden-state dimension on the state DIFFERENCES (:98-108) and one per observation dimension
  * (:110-114), all over the same inputs, keeps (L, alpha) per dimension, and then calls `GpPredictor.computePosterior` with ONE
  * test point for every sigma point, dimension and time step (:72-87) plus once more per dimension for the noise matrices
  * (:138-147): D (or O) launch sequences per call.  Here each family of GPs is ONE resident batch (`DeviceGpFamily`):
  *   - without hyper-parameter optimisation the family is fitted on the device in one call (gp_small_fit: the Y matrix holds one
  *     target column per dimension);
  *   - with it, each dimension's (L, alpha, hyper-parameters) comes from `preComputeComponentsWithHpOptimization` as in the
  *     reference (:119-121; the fit itself is the GpPredictor shim's device L-BFGS) and the batch is assembled from those
  *     factors (gp_small_from_factors);
  *   - a call of the transition / observation / noise function is ONE launch for all dimensions (gp_small_posterior: mean(0)
  *     and sigma(0,0) of every model at the point).
  * Kernels other than GaussianRbfKernel keep the reference's per-dimension computePosterior calls (through the GpPredictor
  * shim, so their O(n^2) work is still on the device). */
class GPUnscentedKalmanFilter(gpOptimizer: GPOptimizer, gpPredictor: GpPredictor) extends UnscentedKalmanFilter(gpOptimizer) {

  import GpPredictor._
  import GPUnscentedKalmanFilter._
  import KalmanFilter._
  import UnscentedKalmanFilter._

  val logger = LoggerFactory.getLogger(classOf[GPUnscentedKalmanFilter])
  val kernelFunc = gpPredictor.kernelFunc
  type aLCWithHyperParams = Array[(afterLearningComponents, Option[KernelFuncHyperParams])]

  def inferHiddenState(input: UnscentedFilteringInput, params: Option[UnscentedTransformParams],
                       computeLL: Boolean, optimizeGpLearning: Boolean): FilteringOutput = {
    val (ukfInput, families, _) = withLearnedModel(input, optimizeGpLearning)
    try inferHiddenState(ukfInput, params, computeLL) finally families.foreach(_.close())
  }

  def inferWithUkfOptimWithWrtToNll(input: UnscentedFilteringInput, initParams: Option[UnscentedTransformParams],
                                    optimizeGpLearning: Boolean, rangeForParam: Range = ukfParamRange) = {
    val (ukfInput, families, hiddenSamples) = withLearnedModel(input, optimizeGpLearning)
    try {
      val objFunction: optimization.Optimization.objectiveFunction = { point: Array[Double] =>
        val out = inferHiddenState(ukfInput, Some(UnscentedTransformParams.fromVector(point)), true)
        val nll = nllOfHiddenData(trueHiddenStates = hiddenSamples, hiddenMeans = out.hiddenMeans, hiddenCovs = out.hiddenCovs)
        if (nll == Double.NegativeInfinity) veryLowValue else if (nll == Double.PositiveInfinity) -veryLowValue else nll
      }
      val (optimizedParams, _) = gpOptimizer.minimize(objFunction, getGpoInput(rangeForParam))
      inferHiddenState(input, Some(UnscentedTransformParams.fromVector(optimizedParams)), true)
    } finally families.foreach(_.close())
  }

  /** :27-35 / :44-49: sample a hidden trajectory from the given model, learn the GP model on it, swap it into the input */
  private def withLearnedModel(input: UnscentedFilteringInput, optimizeGpLearning: Boolean) = {
    val tMax = input.observations.cols
    val initDistr = GaussianDistribution(mean = input.initMean, sigma = input.initCov)
    val (hiddenSamples, _) = input.ssmModel.generateSeries(tMax, SeriesGenerationData(initHiddenState = Right(initDistr)))
    val (gpSsmModel, qNoiseFunc, rNoiseFunc, families) = learnNewSsmModelWithNoises(input.observations, hiddenSamples, optimizeGpLearning)
    (input.copy(ssmModel = gpSsmModel, qNoise = qNoiseFunc, rNoise = rNoiseFunc), families, hiddenSamples)
  }

  private def learnNewSsmModelWithNoises(observations: DenseMatrix[Double], trueHiddenStates: DenseMatrix[Double], optimizeGpLearning: Boolean) = {
    // learnSystemFunction :98-108: inputs = hidden states 0 .. T-2 as rows, targets = differences to the next state
    val systemInput: DenseMatrix[Double] = trueHiddenStates(::, 0 to -2).t.copy
    val transitionDiffs: DenseMatrix[Double] = (trueHiddenStates(::, 1 to -1) - trueHiddenStates(::, 0 to -2)).t.copy   // (T-1) x D
    // learnObsFunction :110-114: inputs = all hidden states as rows, targets = the observations
    val obsInput: DenseMatrix[Double] = trueHiddenStates.t.copy
    val obsTargets: DenseMatrix[Double] = observations.t.copy                                                           // T x O
    val system = learnFamily(systemInput, transitionDiffs, optimizeGpLearning)
    val obs = learnFamily(obsInput, obsTargets, optimizeGpLearning)
    logger.info("Learning of two state functions is done")

    val gpSsmModel: SsmModel = new SsmModel {
      override val transitionFuncImpl: SsmTypeDefinitions.transitionFunc = { (_, prevHiddenState, _) =>
        prevHiddenState + system.meanAt(prevHiddenState)                      // :72-80
      }
      override val observationFuncImpl: SsmTypeDefinitions.observationFunc = { (hiddenState, _) =>
        obs.meanAt(hiddenState)                                               // :81-87
      }
      override val obsNoise: DenseMatrix[Double] = null
      override val latentNoise: DenseMatrix[Double] = null
    }
    val qNoiseFunc: noiseComputationFunc = { context => diag(system.varianceAt(context.hiddenMeans(::, context.iteration - 1))) }          // :89-92
    val rNoiseFunc: noiseComputationFunc = { context => diag(obs.varianceAt(context.firstTransformFromIteration.distribution.mean)) }      // :93-96
    (gpSsmModel, qNoiseFunc, rNoiseFunc, Seq(system, obs))
  }

  /** learnInputOutput (:116-129) for one family: `targets` holds one column per GP */
  private def learnFamily(input: DenseMatrix[Double], targets: DenseMatrix[Double], optimizeGPL: Boolean): GpFamily =
    kernelFunc match {
      case _: GaussianRbfKernel if input.rows <= GPOptimizer.SmallMaxN =>
        if (!optimizeGPL) DeviceGpFamily.fit(input, targets, kernelFunc.hyperParams)
        else {
          val learned = (0 until targets.cols).map(g => gpPredictor.preComputeComponentsWithHpOptimization(input, None, targets(::, g)))
          DeviceGpFamily.fromFactors(input, learned.map(t => (t._1._1, t._1._2, t._2)))
        }
      case _ =>
        val learned: aLCWithHyperParams = (0 until targets.cols).map { g =>
          if (optimizeGPL) { val t = gpPredictor.preComputeComponentsWithHpOptimization(input, None, targets(::, g)); (t._1, Some(t._2)) }
          else (gpPredictor.preComputeComponents(trainingData = input, sigmaNoise = None, targets = targets(::, g)), None)
        }.toArray
        new HostGpFamily(input, learned)
    }

  /** the reference's form: one computePosterior per dimension (:74-77, :140-143) */
  private class HostGpFamily(input: DenseMatrix[Double], learned: aLCWithHyperParams) extends GpFamily {
    private def kf(hp: Option[KernelFuncHyperParams]): KernelFunc = hp.map(h => kernelFunc.changeHyperParams(h.toDenseVector)).getOrElse(kernelFunc)
    private def posterior(point: DenseVector[Double]) = learned.map { case (lc, hp) =>
      gpPredictor.computePosterior(input, point.toDenseMatrix, lc._1, lc._2, kernelFunc = kf(hp))._1
    }
    def meanAt(point: DenseVector[Double]) = DenseVector(posterior(point).map(_.mean(0)))
    def varianceAt(point: DenseVector[Double]) = DenseVector(posterior(point).map(_.sigma(0, 0)))
    def close() {}
  }
}

object GPUnscentedKalmanFilter {

  /** one GP per output dimension over shared inputs: posterior mean / variance of every dimension at ONE point */
  trait GpFamily {
    def meanAt(point: DenseVector[Double]): DenseVector[Double]
    def varianceAt(point: DenseVector[Double]): DenseVector[Double]
    def close(): Unit
  }

  /** a resident gp_small batch: G models, L^-1 and alpha on the device, all dimensions in one launch per point */
  class DeviceGpFamily private (small: Long, g: Int, d: Int) extends GpFamily {
    import Native.{defaultCtx => ctx}
    private def posterior(point: DenseVector[Double]): (Array[Double], Array[Double]) = {
      require(point.length == d)
      val mean = new Array[Double](g); val variance = new Array[Double](g)
      Native.smallPosterior(ctx, small, g, point.toArray, 0, 1, d, 1, mean, variance)      // m = 1: mean[g], var[g]
      (mean, variance)
    }
    def meanAt(point: DenseVector[Double]) = DenseVector(posterior(point)._1)
    def varianceAt(point: DenseVector[Double]) = DenseVector(posterior(point)._2)
    /** (n, capacity, G) of the resident batch */
    def size: (Int, Int, Int) = { val s = Native.smallSize(small); (s(0), s(1), s(2)) }
    /** model `dim`'s (L, alpha) as the reference keeps them per dimension (`afterLearningComponents`, GPUnscentedKalmanFilter.scala:24) */
    def components(dim: Int): (DenseMatrix[Double], DenseVector[Double]) = {
      val n = size._1
      val l = new Array[Double](n * n); val alpha = new Array[Double](n)
      Native.smallGet(ctx, small, 0, dim, l, n)           // GP_SMALL_GET_L
      Native.smallGet(ctx, small, 2, dim, alpha, n)       // GP_SMALL_GET_ALPHA
      (new DenseMatrix(n, n, l), DenseVector(alpha))
    }
    def close() { Native.smallDestroy(small) }
    override def finalize() { close() }
  }

  object DeviceGpFamily {
    import Native.{defaultCtx => ctx, dense, rethrowNotPd}

    /** learnInputOutput without optimisation (:122-124): every dimension with the predictor's own hyper-parameters */
    def fit(input: DenseMatrix[Double], targets: DenseMatrix[Double], hp: KernelFuncHyperParams): DeviceGpFamily = {
      val x = dense(input); val g = targets.cols
      val y = if (targets.isTranspose || targets.majorStride != targets.rows || targets.offset != 0) targets.copy else targets
      val thetas = Array.fill(g)(hp.toDenseVector.toArray).flatten                          // G x (d+2), row-major
      val small = rethrowNotPd { Native.smallFit(ctx, x.data, x.offset, x.rows, x.cols, x.majorStride, y.data, g, thetas, Double.NaN, x.rows) }
      new DeviceGpFamily(small, g, x.cols)
    }

    /** from the (L, alpha, hyper-parameters) that preComputeComponentsWithHpOptimization returned per dimension (:119-121) */
    def fromFactors(input: DenseMatrix[Double], learned: Seq[(DenseMatrix[Double], DenseVector[Double], KernelFuncHyperParams)]): DeviceGpFamily = {
      val x = dense(input); val (n, g) = (x.rows, learned.length)
      val ls = new Array[Double](g * n * n); val alphas = new Array[Double](g * n)
      learned.zipWithIndex.foreach { case ((l, alpha, _), i) =>
        val lc = if (l.isTranspose || l.majorStride != n || l.offset != 0) l.copy else l
        System.arraycopy(lc.data, 0, ls, i * n * n, n * n)
        System.arraycopy(alpha.toArray, 0, alphas, i * n, n)
      }
      val thetas = learned.flatMap(_._3.toDenseVector.toArray).toArray
      val small = Native.smallFromFactors(ctx, x.data, x.offset, n, x.cols, x.majorStride, thetas, g, ls, alphas, n)
      new DeviceGpFamily(small, g, x.cols)
    }
  }
}
