package gp.classification

import breeze.linalg.{DenseMatrix, DenseVector}
import gp.classification.GpClassifier.ClassifierInput
import gpcore.Native
import java.io.{File, PrintWriter}
import org.slf4j.LoggerFactory
import scala.collection.immutable.NumericRange
import utils.KernelRequisites.GaussianRbfKernel

/** Drop-in body for gp.classification.MeshHyperParamsLogLikelihoodEvaluator (MeshHyperParamsLogLikelihoodEvaluator.scala:12-91):
  * the EP log marginal likelihood over a grid of hyper-parameter settings -- SURVEY.md A23 / 8(f) rank 2.
  *
  * The reference walks the grid recursively and runs EP once per visited node, sequentially; as written it evaluates
  * `currentHyperParams` (the array BEFORE the level's value is set, :35-36) but files the result under `copiedHyperParams` (:38),
  * evaluates inner nodes as well as leaves, and keys its map by `Array[Double]` identity, so `getLikelihood` never finds anything
  * (:52-60).  None of that is reproduced (SURVEY.md: "provide a correct batched evaluator; do not replicate the mis-keying"):
  * `evaluate` enumerates the LEAVES of the grid -- every combination of one value per range, last range fastest -- and hands all of
  * them to the library in ONE call (gp_ep_lml_rbf_batched: the settings are identically-shaped problems and run in lockstep on
  * the device, every launch of a sweep covering the whole group); results are kept by VALUE.  `evaluateWithGradients` does the
  * same with the gradient (gp_ep_lml_grad_rbf_batched).  Kernels / stop criteria the device path does not cover run the
  * evaluator's own `logLikelihoodWithoutGrad` per leaf. */
class MeshHyperParamsLogLikelihoodEvaluator(likelihoodEvaluator: MarginalLikelihoodEvaluator) {

  val logger = LoggerFactory.getLogger(this.getClass)

  import MeshHyperParamsLogLikelihoodEvaluator._

  def evaluate(hyperParamsRanges: IndexedSeq[NumericRange[Double]], classificationContext: ClassifierInput): HyperParamsMeshValues = {
    val leaves = meshLeaves(hyperParamsRanges)
    val result = new HyperParamsMeshValues
    val trainData = classificationContext.trainData.get
    val values: Array[Double] = devicePath match {
      case Some(eps) =>
        val thetas = DenseMatrix.tabulate(leaves.length, hyperParamsRanges.length)((b, p) => leaves(b)(p))
        likelihoodEvaluator.logLikelihoodOverMesh(trainData, classificationContext.targets, thetas, eps, HyperParamsOptimization.MaxSweeps).toArray
      case None =>
        leaves.map(theta => likelihoodEvaluator.logLikelihoodWithoutGrad(trainData, classificationContext.targets, DenseVector(theta))).toArray
    }
    leaves.zip(values).foreach { case (theta, likelihood) =>
      logger.info(s"Evaluated likelihood for hyperParams = ${DenseVector(theta)}, value = ${likelihood}")
      result.addResult(theta, likelihood)
    }
    result
  }

  /** (theta, EP log marginal likelihood, its gradient) for every leaf, one library call: the objective a gradient-based fit over
    * many starting points needs (HyperParamsOptimization.scala:38-46 evaluated for a whole population at once). */
  def evaluateWithGradients(hyperParamsRanges: IndexedSeq[NumericRange[Double]], classificationContext: ClassifierInput):
      IndexedSeq[(Array[Double], Double, DenseVector[Double])] = {
    val eps = devicePath.getOrElse(throw new UnsupportedOperationException("batched EP gradient: GaussianRbfKernel with AvgBasedStopCriterion"))
    val leaves = meshLeaves(hyperParamsRanges)
    val (b, p) = (leaves.length, hyperParamsRanges.length)
    val x = Native.dense(classificationContext.trainData.get)
    require(p == x.cols + 2, "one range per hyper-parameter (sf, l_1..l_d, sn)")
    val lml = new Array[Double](b); val grad = new Array[Double](b * p)
    val sweeps = new Array[Int](b); val info = new Array[Int](b)
    Native.epLmlGradRbfBatched(Native.defaultCtx, x.data, x.offset, x.rows, x.cols, x.majorStride, classificationContext.targets.toArray,
      leaves.flatten.toArray, b, eps, HyperParamsOptimization.MaxSweeps, Native.strict, lml, grad, sweeps, info)
    (0 until b).map(i => (leaves(i), lml(i), DenseVector(grad.slice(i * p, (i + 1) * p))))      // NaN where I + S^1/2 K S^1/2 was not PD
  }

  /** Some(eps) when the evaluator's kernel and stop criterion are the ones the batched device path implements */
  private def devicePath: Option[Double] =
    if (likelihoodEvaluator.kernel.isInstanceOf[GaussianRbfKernel]) likelihoodEvaluator.stopEps else None
}

object MeshHyperParamsLogLikelihoodEvaluator {

  import scala.collection._

  /** every combination of one value per range; the first range varies slowest (the order the reference's recursion reaches its leaves in) */
  def meshLeaves(ranges: IndexedSeq[NumericRange[Double]]): IndexedSeq[Array[Double]] =
    ranges.foldLeft(IndexedSeq(Array.empty[Double])) { (prefixes, range) => for (pre <- prefixes; v <- range) yield pre :+ v }

  /** the reference's result holder, keyed by VALUE (immutable.Seq of the setting) so that getLikelihood finds what addResult stored */
  class HyperParamsMeshValues {

    val paramsLikelihood: mutable.LinkedHashMap[immutable.Seq[Double], Double] = mutable.LinkedHashMap()

    def addResult(params: Array[Double], logLikelihood: Double) { paramsLikelihood.put(params.toList, logLikelihood) }

    def getLikelihood(params: Array[Double]): Option[Double] = paramsLikelihood.get(params.toList)

    override def toString: String =
      paramsLikelihood.map { case (params, likelihood) => s"${DenseVector(params.toArray)}  -> ${likelihood}\n" }.mkString

    def writeToFile(fileName: String): Unit = writeToFile(new File(fileName))

    def writeToFile(file: File): Unit = {      // one line per setting: tab-separated hyper-parameters, then the value (:75-89)
      val printWriter = new PrintWriter(file)
      try paramsLikelihood.foreach { case (params, likelihood) => printWriter.write(params.map(v => s"${v}\t").mkString + s"${likelihood}\n") }
      finally printWriter.close()
    }
  }
}
