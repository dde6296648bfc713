package gp.regression

import breeze.linalg.{DenseMatrix, DenseVector}
import gpcore.Native
import gp.regression.Co2Prediction.Co2Kernel
import optimization.Optimization
import optimization.Optimization.BreezeLbfgsOptimizer
import utils.KernelRequisites.{GaussianRbfKernel, KernelFunc, KernelFuncHyperParams}
import utils.MatrixUtils
import utils.StatsUtils.GaussianDistribution

/** Drop-in body for gp.regression.GpPredictor (gp/regression/GpPredictor.scala:15-150): same constructor and method
  * signatures, numerics in libgpcore.so.  Kernel dispatch (SURVEY.md 8b), the same in EVERY method:
  *   GaussianRbfKernel          -> the fused device path (gp_*_rbf);
  *   Co2Prediction.Co2Kernel    -> its own device path (gp_*_co2: Gram, derivatives, fit, LML gradient, L-BFGS);
  *   any other KernelFunc       -> the reference's Scala loops build the matrices (MatrixUtils.buildKernelMatrix /
  *                                 buildMatrixWithFunc) and hand them to gp_fit_from_gram / gp_predict_from_gram /
  *                                 gp_posterior_from_gram / gp_lml_grad_from_gram, so the O(n^3), O(n^2 m) and O(P n^2) work
  *                                 still runs on the GPU; its hyper-parameter fit is the reference's own host L-BFGS
  *                                 (BreezeLbfgsOptimizer) over that objective. */
class GpPredictor(val kernelFunc: KernelFunc) {
  import GpPredictor._
  import Native.{defaultCtx => ctx, dense, rethrowNotPd}

  private def isRbf(k: KernelFunc) = k.isInstanceOf[GaussianRbfKernel]
  private def isCo2(k: KernelFunc) = k.isInstanceOf[Co2Kernel]
  /** the n inputs of a 1-D training / test matrix as the contiguous array the gp_*_co2 entry points take */
  private def column(x: DenseMatrix[Double]): Array[Double] = {
    require(x.cols == 1, "This kernel is applicable only for 1D objects")      // Co2Prediction.scala:40
    x(::, 0).toArray
  }

  private def withModel[T](x: DenseMatrix[Double], hp: KernelFuncHyperParams, sigmaNoise: Option[Double],
                           targets: DenseVector[Double])(body: (Long, KernelFunc) => T): T = {
    require(x.rows == targets.length, "Number of objects in training data matrix should be equal to targets vector length")
    val kf = kernelFunc.changeHyperParams(hp.toDenseVector)
    val model = rethrowNotPd {
      if (isRbf(kf)) {
        val xc = dense(x)   // views cross the API (GpPredictorTest.scala:66): offset and majorStride are passed through
        Native.fitRbf(ctx, xc.data, xc.offset, xc.rows, xc.cols, xc.majorStride, targets.toArray, hp.toDenseVector.toArray,
          sigmaNoise.getOrElse(Double.NaN))
      } else if (isCo2(kf)) {
        Native.fitCo2(ctx, column(x), x.rows, targets.toArray, hp.toDenseVector.toArray, sigmaNoise.getOrElse(Double.NaN))
      } else {
        val k = MatrixUtils.buildKernelMatrix(kf, x)
        sigmaNoise.foreach(v => (0 until k.rows).foreach(i => k(i, i) += v))      // un-squared, GpPredictor.scala:116
        Native.fitFromGram(ctx, k.data, k.offset, k.rows, k.majorStride, targets.toArray)
      }
    }
    try body(model, kf) finally Native.modelDestroy(model)
  }

  def preComputeComponents(trainingData: DenseMatrix[Double], sigmaNoise: Option[Double], targets: DenseVector[Double]): afterLearningComponents =
    preComputeComponents(trainingData, kernelFunc.hyperParams, sigmaNoise, targets)

  def preComputeComponents(trainingData: DenseMatrix[Double], hyperParams: KernelFuncHyperParams,
                           sigmaNoise: Option[Double], targets: DenseVector[Double]): afterLearningComponents =
    withModel(trainingData, hyperParams, sigmaNoise, targets) { (model, _) =>
      val n = trainingData.rows
      val l = new Array[Double](n * n); val alpha = new Array[Double](n)
      Native.modelGet(ctx, model, 0, l, n)
      Native.modelGet(ctx, model, 1, alpha, n)
      (new DenseMatrix(n, n, l), DenseVector(alpha), sigmaNoise.map(v => DenseMatrix.eye[Double](n) :* v))
    }

  def predict(input: PredictionInput, hyperParams: KernelFuncHyperParams = kernelFunc.hyperParams): (GaussianDistribution, Double) =
    withModel(input.trainingData, hyperParams, input.sigmaNoise, input.targets) { (model, kf) =>
      val m = input.testData.rows; val n = input.trainingData.rows
      val mean = new Array[Double](m); val cov = new Array[Double](m * m); val lml = new Array[Double](1)
      if (isRbf(kf) || isCo2(kf)) {      // the model rebuilds K* and K** with its own kernel on the device
        val xs = dense(input.testData)
        Native.predict(ctx, model, xs.data, xs.offset, m, xs.cols, xs.majorStride, mean, null, cov)
      } else {
        val ks = MatrixUtils.buildKernelMatrix(kf, input.testData, input.trainingData)
        val kss = MatrixUtils.buildKernelMatrix(kf, input.testData)
        Native.predictFromGram(ctx, model, ks.data, m, n, kss.data, mean, cov)
      }
      Native.modelGet(ctx, model, 2, lml, 1)
      val fVariance = new DenseMatrix(m, m, cov)
      // GpPredictor.scala:37-39: `fVariance + noiseDiagMtx.get` adds the n x n noise matrix to the m x m covariance, so
      // with sigmaNoise = Some(_) the reference only works for m == n; Breeze's own dimension check is kept by doing the
      // same addition
      val withNoise = input.sigmaNoise match {
        case Some(v) => fVariance + (DenseMatrix.eye[Double](n) :* v)
        case None => fVariance
      }
      (GaussianDistribution(mean = DenseVector(mean), sigma = withNoise), lml(0))
    }

  def computePosterior(trainingData: DenseMatrix[Double], testData: DenseMatrix[Double], l: DenseMatrix[Double],
                       alphaVec: DenseVector[Double]): (GaussianDistribution, DenseMatrix[Double]) =
    computePosterior(trainingData, testData, l, alphaVec, kernelFunc)

  // GpPredictor.scala:50-58 -- the entry GPOptimizer.scala:91 and GPUnscentedKalmanFilter.scala:78-87,141-142 call
  def computePosterior(trainingData: DenseMatrix[Double], testData: DenseMatrix[Double], l: DenseMatrix[Double],
                       alphaVec: DenseVector[Double], kernelFunc: KernelFunc): (GaussianDistribution, DenseMatrix[Double]) = {
    val n = trainingData.rows; val m = testData.rows
    val mean = new Array[Double](m); val cov = new Array[Double](m * m); val v = new Array[Double](n * m)
    val lc = dense(l)
    kernelFunc match {
      case rbf: GaussianRbfKernel =>
        val x = dense(trainingData); val xs = dense(testData)
        Native.posteriorFromFactor(ctx, x.data, x.offset, n, x.cols, x.majorStride, rbf.hyperParams.toDenseVector.toArray,
          lc.data, lc.offset, lc.majorStride, alphaVec.toArray, xs.data, xs.offset, m, xs.majorStride, mean, null, cov, v)
      case co2: Co2Kernel =>          // K* and K** on the device as well (gp_cross_gram_co2 / gp_gram_co2), then the same solves
        val th = co2.hyperParams.toDenseVector.toArray
        val ks = new Array[Double](m * n); val kss = new Array[Double](m * m)
        Native.crossGramCo2(ctx, column(testData), m, column(trainingData), n, th, ks)
        Native.gramCo2(ctx, column(testData), m, th, kss)
        Native.posteriorFromGram(ctx, ks, m, n, kss, lc.data, lc.offset, lc.majorStride, alphaVec.toArray, mean, cov, v)
      case other =>
        val ks = MatrixUtils.buildKernelMatrix(other, testData, trainingData)
        val kss = MatrixUtils.buildKernelMatrix(other, testData)
        Native.posteriorFromGram(ctx, ks.data, m, n, kss.data, lc.data, lc.offset, lc.majorStride, alphaVec.toArray, mean, cov, v)
    }
    (GaussianDistribution(mean = DenseVector(mean), sigma = new DenseMatrix(m, m, cov)), new DenseMatrix(n, m, v))
  }

  def logLikelihoodWithDerivatives(input: PredictionTrainingInput, hyperParams: KernelFuncHyperParams,
                                   optimizedParamsNum: Int): (Double, DenseVector[Double]) = {
    val kf = kernelFunc.changeHyperParams(hyperParams.toDenseVector)
    val theta = hyperParams.toDenseVector.toArray
    val sn = input.sigmaNoise.getOrElse(Double.NaN)
    val lml = new Array[Double](1); val grad = new Array[Double](optimizedParamsNum); val info = new Array[Int](1)
    kf match {
      case _: GaussianRbfKernel =>
        val x = dense(input.trainingData)
        Native.lmlGradBatched(ctx, x.data, x.offset, x.rows, x.cols, x.majorStride, input.targets.toArray, theta, 1,
          optimizedParamsNum, sn, lml, grad, info)
      case _: Co2Kernel =>
        Native.lmlGradCo2Batched(ctx, column(input.trainingData), input.trainingData.rows, input.targets.toArray, theta, 1,
          optimizedParamsNum, sn, lml, grad, info)
      case other =>
        // GpPredictor.scala:62,74 on the host (the kernel is opaque Scala code), :63-78 on the device
        val k = MatrixUtils.buildKernelMatrix(other, input.trainingData)
        val dks = (1 to optimizedParamsNum).map { i =>
          MatrixUtils.buildMatrixWithFunc(input.trainingData)(other.derAfterHyperParam(i)).data }.toArray
        rethrowNotPd {
          lml(0) = Native.lmlGradFromGram(ctx, k.data, k.offset, k.rows, k.majorStride, input.targets.toArray, dks, k.rows, sn, grad)
        }
    }
    if (info(0) != 0)   // breeze.linalg.cholesky throws inside preComputeComponents (GpPredictor.scala:63-64,120)
      throw new breeze.linalg.NotConvergedException(breeze.linalg.NotConvergedException.Iterations,
        "matrix not positive definite at pivot " + info(0))
    (lml(0), DenseVector(grad))
  }

  // GpPredictor.scala:126-142 -- BreezeLbfgsOptimizer(maxIter = 20), m = 4, best-seen point; optimizeNoise = false keeps sn fixed
  def obtainOptimalHyperParams(trainingData: DenseMatrix[Double], sigmaNoise: Option[Double], targets: DenseVector[Double],
                               optimizeNoise: Boolean): KernelFuncHyperParams = {
    val theta = kernelFunc.hyperParams.toDenseVector.toArray
    val nparams = if (optimizeNoise) theta.length else theta.length - 1
    val sn = sigmaNoise.getOrElse(Double.NaN)
    kernelFunc match {
      case _: GaussianRbfKernel =>     // device-resident L-BFGS, trial steps of a line search as one lockstep batch
        val x = dense(trainingData)
        rethrowNotPd { Native.optimizeRbf(ctx, x.data, x.offset, x.rows, x.cols, x.majorStride, targets.toArray, theta, nparams, sn, 20, 4) }
        kernelFunc.hyperParams.fromDenseVector(DenseVector(theta))
      case _: Co2Kernel =>             // MasterThesisRelatedTasks.scala:59-71 -> predictWithParamsOptimization on co2GpPredictor
        rethrowNotPd { Native.optimizeCo2(ctx, column(trainingData), trainingData.rows, targets.toArray, theta, nparams, sn, 20, 4) }
        kernelFunc.hyperParams.fromDenseVector(DenseVector(theta))
      case _ =>                        // the reference's own loop (:128-141): host L-BFGS over the (device-evaluated) objective
        val breezeOptimizer = new BreezeLbfgsOptimizer(maxIter = 20)
        val initPoint = if (optimizeNoise) theta else theta.take(theta.length - 1)
        val llObjFunction: Optimization.objectiveFunctionWithGradient = { currentParams =>
          val hyperParams = kernelFunc.hyperParams.fromDenseVector(DenseVector(currentParams))
          val ptInput = PredictionTrainingInput(trainingData = trainingData, targets = targets, sigmaNoise = sigmaNoise)
          val (value, gradient) = logLikelihoodWithDerivatives(ptInput, hyperParams, currentParams.length)
          (value, gradient.toArray)
        }
        kernelFunc.hyperParams.fromDenseVector(DenseVector(breezeOptimizer.maximize(llObjFunction, initPoint)))
    }
  }

  def predictWithParamsOptimization(input: PredictionInput, optimizeNoise: Boolean): (GaussianDistribution, Double, KernelFuncHyperParams) = {
    val hp = obtainOptimalHyperParams(input.trainingData, input.sigmaNoise, input.targets, optimizeNoise)
    val (dist, ll) = predict(input, hyperParams = hp)
    (dist, ll, hp)
  }

  def preComputeComponentsWithHpOptimization(trainingData: DenseMatrix[Double], sigmaNoise: Option[Double],
                                             targets: DenseVector[Double]): (afterLearningComponents, KernelFuncHyperParams) = {
    val hp = obtainOptimalHyperParams(trainingData, sigmaNoise, targets, true)
    (preComputeComponents(trainingData, hp, sigmaNoise, targets), hp)
  }
}

object GpPredictor {
  type afterLearningComponents = (DenseMatrix[Double], DenseVector[Double], Option[DenseMatrix[Double]])
  case class PredictionInput(trainingData: DenseMatrix[Double], testData: DenseMatrix[Double],
                             sigmaNoise: Option[Double], targets: DenseVector[Double]) {
    def toPredictionTrainingInput: PredictionTrainingInput =
      PredictionTrainingInput(trainingData = trainingData, sigmaNoise = sigmaNoise, targets = targets)
  }
  case class PredictionTrainingInput(trainingData: DenseMatrix[Double], sigmaNoise: Option[Double], targets: DenseVector[Double])
}
