package gp.regression

import breeze.linalg.{DenseMatrix, DenseVector}
import gpcore.Native
import utils.KernelRequisites.{GaussianRbfKernel, KernelFunc, KernelFuncHyperParams}
import utils.StatsUtils.GaussianDistribution

/** Drop-in body for gp.regression.GpPredictor: same constructor and method signatures as the reference class,
  * numerics in libgpcore.so.  A GaussianRbfKernel goes to the fused device path; any other KernelFunc keeps the
  * reference's Scala loops for the Gram matrix and hands it over through gp_fit_from_gram (not shown). */
class GpPredictor(val kernelFunc: KernelFunc) {
  import GpPredictor._

  private def compact(m: DenseMatrix[Double]): DenseMatrix[Double] =
    if (m.isTranspose || m.majorStride != m.rows) m.copy else m      // views cross the API (GpPredictorTest.scala:66)

  private def withModel[T](x: DenseMatrix[Double], hp: KernelFuncHyperParams, sigmaNoise: Option[Double],
                           targets: DenseVector[Double])(body: Long => T): T = {
    require(x.rows == targets.length, "Number of objects in training data matrix should be equal to targets vector length")
    val xc = compact(x)
    val model = Native.fitRbf(Native.defaultCtx, xc.data, xc.offset, xc.rows, xc.cols, xc.majorStride,
      targets.toArray, hp.toDenseVector.toArray, sigmaNoise.getOrElse(Double.NaN))
    try body(model) finally Native.modelDestroy(model)
  }

  def preComputeComponents(trainingData: DenseMatrix[Double], hyperParams: KernelFuncHyperParams,
                           sigmaNoise: Option[Double], targets: DenseVector[Double]): afterLearningComponents =
    withModel(trainingData, hyperParams, sigmaNoise, targets) { model =>
      val n = trainingData.rows
      val l = new Array[Double](n * n); val alpha = new Array[Double](n)
      Native.modelGet(Native.defaultCtx, model, 0, l, n)
      Native.modelGet(Native.defaultCtx, model, 1, alpha, n)
      (new DenseMatrix(n, n, l), DenseVector(alpha), sigmaNoise.map(v => DenseMatrix.eye[Double](n) :* v))
    }

  def predict(input: PredictionInput, hyperParams: KernelFuncHyperParams = kernelFunc.hyperParams): (GaussianDistribution, Double) =
    withModel(input.trainingData, hyperParams, input.sigmaNoise, input.targets) { model =>
      val xs = compact(input.testData); val m = xs.rows
      val mean = new Array[Double](m); val cov = new Array[Double](m * m); val lml = new Array[Double](1)
      Native.predict(Native.defaultCtx, model, xs.data, xs.offset, m, xs.majorStride, mean, null, cov)
      Native.modelGet(Native.defaultCtx, model, 2, lml, 1)
      (GaussianDistribution(mean = DenseVector(mean), sigma = new DenseMatrix(m, m, cov)), lml(0))
    }

  def logLikelihoodWithDerivatives(input: PredictionTrainingInput, hyperParams: KernelFuncHyperParams,
                                   optimizedParamsNum: Int): (Double, DenseVector[Double]) = {
    val x = compact(input.trainingData)
    val lml = new Array[Double](1); val grad = new Array[Double](optimizedParamsNum); val info = new Array[Int](1)
    Native.lmlGradBatched(Native.defaultCtx, x.data, x.rows, x.cols, x.majorStride, input.targets.toArray,
      hyperParams.toDenseVector.toArray, 1, optimizedParamsNum, input.sigmaNoise.getOrElse(Double.NaN), lml, grad, info)
    (lml(0), DenseVector(grad))
  }

  // GpPredictor.scala:126-142 -- BreezeLbfgsOptimizer(maxIter = 20), m = 4, best-seen point; optimizeNoise = false keeps sn fixed
  def obtainOptimalHyperParams(trainingData: DenseMatrix[Double], sigmaNoise: Option[Double], targets: DenseVector[Double],
                               optimizeNoise: Boolean): KernelFuncHyperParams = {
    val x = compact(trainingData)
    val theta = kernelFunc.hyperParams.toDenseVector.toArray
    Native.optimizeRbf(Native.defaultCtx, x.data, x.rows, x.cols, x.majorStride, targets.toArray, theta,
      if (optimizeNoise) theta.length else theta.length - 1, sigmaNoise.getOrElse(Double.NaN), 20, 4)
    kernelFunc.hyperParams.fromDenseVector(DenseVector(theta))
  }
}

object GpPredictor {
  type afterLearningComponents = (DenseMatrix[Double], DenseVector[Double], Option[DenseMatrix[Double]])
  case class PredictionInput(trainingData: DenseMatrix[Double], testData: DenseMatrix[Double],
                             sigmaNoise: Option[Double], targets: DenseVector[Double])
  case class PredictionTrainingInput(trainingData: DenseMatrix[Double], sigmaNoise: Option[Double], targets: DenseVector[Double])
}
