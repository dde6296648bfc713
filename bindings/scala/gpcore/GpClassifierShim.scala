package gp.classification

import breeze.linalg.{DenseMatrix, DenseVector}
import gp.classification.EpParameterEstimator.SiteParams
import gpcore.Native
import utils.KernelRequisites.KernelFuncHyperParams

/** Drop-in body for gp.classification.GpClassifier (gp/classification/GpClassifier.scala:11-67). */
class GpClassifier(stopCriterion: EpParameterEstimator.stopCriterionFunc) {
  import GpClassifier._
  import Native.{defaultCtx => ctx, dense, rethrowNotPd}

  /* targets must contain values from set {-1,1} */
  def trainClassifier(classInput: ClassifierInput): learnParams =
    new EpParameterEstimator(classInput.trainKernelMatrix, classInput.targets, stopCriterion).estimateSiteParams

  // :24-47 -- z, fMean, V = L \ (S^1/2 K*^T), p_i = Phi(mu*_i / sqrt(1 + var*_i)) in ONE library call; only the diagonal of
  // testKernelMatrix is read by the reference (:43-45), so only the diagonal crosses the boundary
  def classify(input: AfterEstimationClassifierInput): classifyOut = {
    val (siteParams, _) = input.learnParams.getOrElse(
      trainClassifier(ClassifierInput(trainKernelMatrix = input.trainKernelMatrix, targets = input.targets,
        initHyperParams = input.hyperParams, trainData = None)))
    val k = dense(input.trainKernelMatrix); val ks = dense(input.testTrainKernelMatrix)
    val n = k.rows; val m = ks.rows
    val ep = Native.epCreate(ctx, k.data, k.offset, n, k.majorStride, input.targets.toArray)
    try {
      rethrowNotPd { Native.epSetSiteParams(ctx, ep, n, siteParams.tauSiteParams.toArray, siteParams.niSiteParams.toArray) }
      val kssDiag = Array.tabulate(m)(i => input.testKernelMatrix(i, i))
      val prob = new Array[Double](m)
      Native.epPredict(ctx, ep, ks.data, ks.offset, m, n, ks.majorStride, kssDiag, prob)
      DenseVector(prob)
    } finally Native.epDestroy(ep)
  }
}

object GpClassifier {
  type classifyOut = DenseVector[Double]
  /* second elem from tuple is an output from cholesky decomposition */
  type learnParams = (SiteParams, DenseMatrix[Double])

  case class AfterEstimationClassifierInput(targets: DenseVector[Int], learnParams: Option[learnParams],
                                            hyperParams: KernelFuncHyperParams, trainKernelMatrix: DenseMatrix[Double],
                                            testTrainKernelMatrix: DenseMatrix[Double], testKernelMatrix: DenseMatrix[Double])
  case class ClassifierInput(trainKernelMatrix: DenseMatrix[Double], targets: DenseVector[Int],
                             initHyperParams: KernelFuncHyperParams, trainData: Option[DenseMatrix[Double]])
}
