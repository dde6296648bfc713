package gp.classification

import breeze.linalg.{DenseMatrix, DenseVector}
import gp.classification.EpParameterEstimator.AvgBasedStopCriterion
import gp.classification.GpClassifier.ClassifierInput
import gp.optimization.GPOptimizer
import gpcore.Native
import org.slf4j.LoggerFactory
import utils.KernelRequisites.{GaussianRbfKernel, KernelFuncHyperParams}

/** Drop-in body for gp.classification.HyperParamsOptimization (gp/classification/HyperParamsOptimization.scala:19-140):
  * hyper-parameter fitting of the EP classifier by maximising the EP log marginal likelihood -- SURVEY.md 8(f) rank 2.
  *
  * `GradientHyperParamsOptimizer(marginalLikelihoodEvaluator, gradOptimizer)` keeps its constructor and `optimizeHyperParams`.
  * With the pieces the reference itself wires together (GaussianRbfKernel, BreezeLbfgsOptimizer, AvgBasedStopCriterion:
  * resources/config/spring-context.xml:33-59) the WHOLE optimisation is one library call (gp_ep_optimize_rbf: Gram matrix, EP
  * runs, gradient and L-BFGS on the device, the trial steps of a line search as one lockstep batch) instead of one EP run with
  * two factorisations per objective call (:38-46).  Anything else -- another kernel, another optimiser, a stop criterion that
  * is an arbitrary function -- keeps the reference's loop: `gradOptimizer.maximize` over `marginalLikelihoodEvaluator.logLikelihood`,
  * whose EP run and gradient are the MarginalLikelihoodEvaluator shim's (device) ones.
  *
  * `ApacheCommonsOptimizer` (:57-136) is NOT shimmed: it only ever calls `marginalLikelihoodEvaluator.logLikelihood`, so the
  * reference's class runs unchanged on top of the shimmed evaluator; keep the reference's text for it when this file replaces
  * the original. */
object HyperParamsOptimization {

  import optimization.Optimization._

  val apacheLogger = LoggerFactory.getLogger(classOf[HyperParamsOptimization.HyperParameterOptimizer])

  trait HyperParameterOptimizer {
    def optimizeHyperParams(optimizationInput: ClassifierInput): KernelFuncHyperParams
  }

  /** What the device-resident fit needs to know about the evaluator: its kernel and, if its stop criterion is the reference's
    * AvgBasedStopCriterion, that criterion's eps.  The MarginalLikelihoodEvaluator shim exposes both (`kernelFunc`, `stopEps`). */
  class GradientHyperParamsOptimizer(marginalLikelihoodEvaluator: MarginalLikelihoodEvaluator, gradOptimizer: GradientBasedOptimizer)
    extends HyperParameterOptimizer {

    def optimizeHyperParams(optimizationInput: ClassifierInput): KernelFuncHyperParams = {
      val targets = optimizationInput.targets
      val trainData = optimizationInput.trainData.get
      val init = optimizationInput.initHyperParams
      val native = for {
        maxIter <- GPOptimizer.lbfgsMaxIter(gradOptimizer)                               // the reference's BreezeLbfgsOptimizer
        eps <- marginalLikelihoodEvaluator.stopEps                                       // AvgBasedStopCriterion(eps)
        if marginalLikelihoodEvaluator.kernel.isInstanceOf[GaussianRbfKernel]
      } yield (maxIter, eps)
      native match {
        case Some((maxIter, eps)) =>
          val x = Native.dense(trainData)
          val theta = init.toDenseVector.toArray                                         // in: start, out: best point seen
          val best = Native.rethrowNotPd {
            Native.epOptimizeRbf(Native.defaultCtx, x.data, x.offset, x.rows, x.cols, x.majorStride, targets.toArray, theta, eps,
              MaxSweeps, Native.strict, maxIter, 4)                                      // L-BFGS m = 4 (Optimization.scala:34-35)
          }
          apacheLogger.info(s"Optimal solution is = ${DenseVector(theta)}, objective function value = ${best}")
          init.fromDenseVector(DenseVector(theta))
        case None =>
          val funcWithGradient: objectiveFunctionWithGradient = { hyperParams: Array[Double] =>
            val (logLikelihood, derivatives) = marginalLikelihoodEvaluator.logLikelihood(trainData, targets, DenseVector(hyperParams))
            assert(hyperParams.length == derivatives.length)
            (logLikelihood, derivatives.toArray)
          }
          init.fromDenseVector(DenseVector(gradOptimizer.maximize(funcWithGradient, init.toDenseVector.toArray)))
      }
    }
  }

  /** sweeps after which an EP run that has not met its criterion is cut (the reference loops without a bound, EpParameterEstimator.scala:40) */
  val MaxSweeps = 1000

  case class HyperOptimizationInput(trainInput: DenseMatrix[Double], targets: DenseVector[Int], initParams: KernelFuncHyperParams)
}
