package gp.optimization

import breeze.linalg.{DenseMatrix, DenseVector}
import gp.regression.GpPredictor
import gpcore.Native
import optimization.Optimization._
import scala.beans.BeanProperty
import scala.util.Random
import utils.KernelRequisites.{GaussianRbfKernel, KernelFuncHyperParams}
import utils.StatsUtils._

/** Drop-in body for gp.optimization.GPOptimizer (gp/optimization/GPOptimizer.scala:19-172): GP-UCB Bayesian optimisation.
  * Same constructor, `maximize` / `minimize` / `prepareGrid` / `evaluateGridPoints` and `GPOInput`; what changes is how an
  * iteration runs (SURVEY.md 8(f) rank 3):
  *   - the model is fitted ONCE on the initial grid (gp_small_fit: L, L^-1 and alpha resident on the device) and every chosen
  *     point is APPENDED by a rank-1 extension of L, L^-1 and alpha (gp_small_append, O(n^2)) where the reference refits from
  *     scratch in every iteration (`preComputeComponents`, :51) and inverts L again in every `maximizeUCB` (:85);
  *   - with the reference's own BreezeLbfgsOptimizer the c L-BFGS runs of one iteration (:55-63) advance in lockstep, all trial
  *     points of an L-BFGS iteration in one launch (gp_small_maximize_ucb: objective mean + k sqrt(var) and its gradient
  *     through GaussianRbfKernel.gradient, utils/KernelRequisites.scala:95-107, on the device);
  *   - any other GradientBasedOptimizer keeps driving the search itself, one start at a time as in the reference, with the
  *     objective of :88-106 evaluated by ONE launch per trial point (gp_small_ucb) instead of computePosterior + two derivative
  *     matrices + a product with L^-1 on the host.
  * `KernelFunc.gradient` is implemented by GaussianRbfKernel only (Co2Kernel's is `???`, Co2Prediction.scala:62-64), so that is
  * the kernel this class has ever worked with; anything else fails the `require` below instead of the reference's NotImplementedError. */
class GPOptimizer(@BeanProperty var gpPredictor: GpPredictor, noise: Option[Double], gradientOptimizer: GradientBasedOptimizer) {
  import GPOptimizer._
  import Native.{defaultCtx => ctx, dense, rethrowNotPd}

  val hyperParams = gpPredictor.kernelFunc.hyperParams

  def minimize(objFunc: objectiveFunction, params: GPOInput): (Array[Double], Double) = {
    val (optimum, optimumValue) = maximize({ point: Array[Double] => -objFunc(point) }, params)
    (optimum, -optimumValue)
  }

  def maximize(func: objectiveFunction, params: GPOInput): (Array[Double], Double) = {
    val (ranges, m, c, k) = (params.ranges, params.mParam, params.cParam, params.kParam)
    require(c >= 1 && m >= 1, "Params m and c needs to be greater or equal 1")
    require(gpPredictor.kernelFunc.isInstanceOf[GaussianRbfKernel], "GP-UCB needs KernelFunc.gradient: GaussianRbfKernel")
    val pointGrid = prepareGrid(ranges)
    val evaluatedPointGrid = evaluateGridPoints(pointGrid, func)
    val hp: KernelFuncHyperParams =
      if (params.optimizeHpOnInitGrid) gpPredictor.obtainOptimalHyperParams(pointGrid, noise, evaluatedPointGrid, true) else hyperParams
    val (n0, d) = (pointGrid.rows, pointGrid.cols)
    require(n0 + m <= SmallMaxN, s"3 d + m = ${n0 + m} exceeds GP_SMALL_MAX_N = $SmallMaxN")

    // the point set and its values grow on the host exactly as in the reference (:47-72); the model grows on the device
    var points = Vector.tabulate(n0)(i => pointGrid(i, ::).t.toArray)
    var values = evaluatedPointGrid.toArray.toVector
    val x = dense(pointGrid)
    val small = rethrowNotPd {
      Native.smallFit(ctx, x.data, x.offset, n0, d, x.majorStride, evaluatedPointGrid.toArray, 1, hp.toDenseVector.toArray,
        noise.getOrElse(Double.NaN), n0 + m)
    }
    try {
      for (iterNum <- 0 until m) {
        val asMatrix = DenseMatrix.tabulate(points.length, d)((i, j) => points(i)(j))
        val (observedMean, observedCov) = meanAndVarOfData(asMatrix)                        // :52
        val sampler = new NormalDistributionSampler(GaussianDistribution(mean = observedMean, sigma = observedCov))
        val starts = Array.fill(c)(sampler.sample.toArray)                                  // :57, one start per L-BFGS run
        val (biggestUcb, found) = maximizeUcbOnDevice(small, starts, d, k)
        val biggestUcbPoint = if (found.isEmpty || biggestUcb.isNaN) sampler.sample.toArray else found.get   // :64
        try {
          val evaluated = func(biggestUcbPoint)                                             // :66-69
          rethrowNotPd { Native.smallAppend(ctx, small, d, 1, biggestUcbPoint, Array(evaluated)) }
          points = points :+ biggestUcbPoint
          values = values :+ evaluated
        } catch {
          case _: Exception =>                                                              // :70-72: the sets stay as they were
        }
      }
    } finally Native.smallDestroy(small)
    val maxIndex = values.indices.foldLeft(0)((best, i) => if (values(i) > values(best)) i else best)   // first maximum, :73-78
    (points(maxIndex), values(maxIndex))
  }

  /** The c runs of maximizeUCB (:55-63, 82-109) of one iteration against the resident model: (best UCB, its point). */
  private def maximizeUcbOnDevice(small: Long, starts: Array[Array[Double]], d: Int, kParam: Double): (Double, Option[Array[Double]]) =
    lbfgsMaxIter(gradientOptimizer) match {
      case Some(maxIter) =>      // the reference's optimiser: all c runs in lockstep on the device (L-BFGS m = 4, Optimization.scala:34-35)
        val c = starts.length
        val colMajor = Array.tabulate(c * d)(idx => starts(idx % c)(idx / c))               // c x d, column-major
        val bestX = new Array[Double](d)
        val best = Native.smallMaximizeUcb(ctx, small, 0, colMajor, c, d, kParam, maxIter, 4, bestX)
        (best, Some(bestX))
      case None =>               // somebody else's optimiser drives: one launch per trial point
        val objective: objectiveFunctionWithGradient = { testPoint =>
          val value = new Array[Double](1); val grad = new Array[Double](d)
          Native.smallUcb(ctx, small, 0, testPoint, 0, 1, d, 1, kParam, value, grad)
          (value(0), grad)
        }
        starts.foldLeft[(Double, Option[Array[Double]])]((Double.MinValue, None)) { case ((bestVal, bestPoint), start) =>
          val optimum = gradientOptimizer.maximize(objective, start)
          val ucbValue = objective(optimum)._1
          if (ucbValue > bestVal) (ucbValue, Some(optimum)) else (bestVal, bestPoint)
        }
    }

  /*grid(i,::) - i'th d-dimensional point*/
  def evaluateGridPoints(grid: DenseMatrix[Double], func: objectiveFunction): DenseVector[Double] =
    DenseVector.tabulate(grid.rows)(i => func(grid(i, ::).t.toArray))

  def prepareGrid(ranges: IndexedSeq[Range]): DenseMatrix[Double] = {       // :138-152: 3 dim uniform points inside the ranges
    val rand = new Random(System.nanoTime())
    DenseMatrix.tabulate(3 * ranges.length, ranges.length) { (_, j) =>
      val (lower, upper) = (ranges(j).start.toDouble, ranges(j).end.toDouble)
      require(lower < upper)
      lower + (upper - lower) * rand.nextDouble()
    }
  }
}

object GPOptimizer {

  case class GPOInput(ranges: IndexedSeq[Range], mParam: Int, cParam: Int, kParam: Double, optimizeHpOnInitGrid: Boolean = false)

  /** GP_SMALL_MAX_N of include/gpcore.h */
  val SmallMaxN = 2048

  /** `maxIter` of a BreezeLbfgsOptimizer (a constructor parameter the class keeps as a private field for its factory closure,
    * optimization/Optimization.scala:30-35; default constructor: 10); None for any other optimiser. */
  def lbfgsMaxIter(opt: GradientBasedOptimizer): Option[Int] = opt match {
    case b: BreezeLbfgsOptimizer =>
      try {
        val f = classOf[BreezeLbfgsOptimizer].getDeclaredFields.find(_.getName.endsWith("maxIter")).get
        f.setAccessible(true)
        Some(f.getInt(b))
      } catch { case _: Exception => Some(10) }
    case _ => None
  }
}
