package utils

import breeze.linalg.{DenseMatrix, DenseVector}
import gpcore.Native

/** Drop-in bodies for object utils.MatrixUtils (utils/MatrixUtils.scala:17-133): same names and signatures.  ARD-RBF Gram
  * matrices and every triangular solve run in libgpcore.so; an arbitrary KernelFunc or 3-argument function keeps the
  * reference's own loops (they are the definition of that kernel). */
object MatrixUtils {
  import KernelRequisites._
  import Native.{defaultCtx => ctx, dense}

  type rowMatrixRange = Int => Range

  // ---- triangular solves: forwardSolve(L, b) = L \ b, backSolve(R, b) = R \ b with R UPPER (callers pass L.t) -- :17-35
  private def solve(trans: Int, lower: DenseMatrix[Double], b: DenseMatrix[Double]): DenseMatrix[Double] = {
    require(lower.rows == lower.cols)                                   // solveTriangular :125
    val l = dense(lower); val x = b.copy                                // result is a fresh matrix, like the reference's
    Native.trsmLower(ctx, trans, l.data, l.offset, l.rows, l.majorStride, x.data, x.offset, x.cols, x.majorStride)
    x
  }
  def forwardSolve(L: DenseMatrix[Double], b: DenseMatrix[Double]): DenseMatrix[Double] = solve(0, L, b)
  def backSolve(R: DenseMatrix[Double], b: DenseMatrix[Double]): DenseMatrix[Double] = solve(1, R.t, b)   // R = L.t  =>  R.t = L
  def forwardSolve(L: DenseMatrix[Double], b: DenseVector[Double]): DenseVector[Double] =
    forwardSolve(L, b.toDenseMatrix.t)(::, 0)
  def backSolve(R: DenseMatrix[Double], b: DenseVector[Double]): DenseVector[Double] =
    backSolve(R, b.toDenseMatrix.t)(::, 0)

  def invTriangular(matrix: DenseMatrix[Double], isUpper: Boolean): DenseMatrix[Double] = {   // :106-113
    val n = matrix.rows
    if (isUpper) backSolve(R = matrix, b = DenseMatrix.eye[Double](n))
    else {
      val l = dense(matrix); val out = new Array[Double](n * n)
      Native.invLower(ctx, l.data, l.offset, n, l.majorStride, out)
      new DenseMatrix(n, n, out)
    }
  }

  def cloneCols(vec: DenseVector[Double], colNum: Int): DenseMatrix[Double] = {   // :37-42
    val result = DenseMatrix.zeros[Double](vec.length, colNum)
    (0 until colNum).foreach(c => result(::, c) := vec)
    result
  }

  // ---- Gram builders :44-97
  def buildKernelMatrix(kernelFun: KernelFunc, input1: DenseMatrix[Double], input2: DenseMatrix[Double]): kernelMatrixType =
    kernelFun match {
      case rbf: GaussianRbfKernel =>
        val a = dense(input1); val b = dense(input2); val out = new Array[Double](a.rows * b.rows)
        Native.crossGramRbf(ctx, a.data, a.offset, a.rows, a.majorStride, b.data, b.offset, b.rows, b.majorStride, a.cols,
          rbf.hyperParams.toDenseVector.toArray, out)
        new DenseMatrix(a.rows, b.rows, out)
      case _ => buildMatrixWithFunc({ (v1, v2) => kernelFun.apply(v1, v2, false) }, input1, input2)
    }

  def buildKernelMatrix(kernelFun: KernelFunc, data: DenseMatrix[Double]): kernelMatrixType =
    kernelFun match {
      case rbf: GaussianRbfKernel =>
        val x = dense(data); val n = x.rows; val out = new Array[Double](n * n)
        Native.gramRbf(ctx, x.data, x.offset, n, x.cols, x.majorStride, rbf.hyperParams.toDenseVector.toArray, out)
        new DenseMatrix(n, n, out)
      case _ => buildMatrixWithFunc(data)((v1, v2, same) => kernelFun.apply(v1, v2, same))
    }

  /** dK/dtheta_pos for the ARD-RBF kernel on the device (what buildMatrixWithFunc(X)(kernel.derAfterHyperParam(pos)) builds) */
  def buildRbfDerivativeMatrix(rbf: GaussianRbfKernel, data: DenseMatrix[Double], pos: Int): kernelMatrixType = {
    val x = dense(data); val n = x.rows; val out = new Array[Double](n * n)
    Native.dgramRbf(ctx, x.data, x.offset, n, x.cols, x.majorStride, rbf.hyperParams.toDenseVector.toArray, pos, out)
    new DenseMatrix(n, n, out)
  }

  /** dK/dhp_pos of gp.regression.Co2Prediction.Co2Kernel (Co2Prediction.scala:66-137) over 1-D inputs, on the device: what
    * buildMatrixWithFunc(X)(co2Kernel.derAfterHyperParam(pos)) builds; `theta` = hp1..hp11 */
  def buildCo2DerivativeMatrix(theta: DenseVector[Double], data: DenseMatrix[Double], pos: Int): kernelMatrixType = {
    require(data.cols == 1, "This kernel is applicable only for 1D objects")
    val n = data.rows; val out = new Array[Double](n * n)
    Native.dgramCo2(ctx, data(::, 0).toArray, n, theta.toArray, pos, out)
    new DenseMatrix(n, n, out)
  }

  /** breeze.linalg.cholesky(A) for the reference's own call sites that hold a ready-made matrix (GpPredictor.scala:120,
    * EpParameterEstimator.scala:58, UnscentedKalmanFilter.scala:90): lower factor with a zero upper triangle, on the device;
    * a non-positive pivot throws what Breeze throws. */
  def choleskyLower(a: DenseMatrix[Double]): DenseMatrix[Double] = {
    require(a.rows == a.cols)
    val l = a.copy
    Native.rethrowNotPd { Native.potrfLower(ctx, l.data, l.offset, l.rows, l.majorStride) }
    l
  }

  def buildMatrixWithFunc(data: DenseMatrix[Double])(f: (DenseVector[Double], DenseVector[Double], Boolean) => Double): kernelMatrixType = {
    val n = data.rows
    val result = DenseMatrix.zeros[Double](n, n)
    for (i <- 0 until n; j <- 0 to i) {
      val value = f(data(i, ::).t, data(j, ::).t, i == j)
      result.update(i, j, value); result.update(j, i, value)
    }
    result
  }

  def buildMatrixWithFunc(func: (DenseVector[Double], DenseVector[Double]) => Double, input1: DenseMatrix[Double],
                          input2: DenseMatrix[Double]): kernelMatrixType = {
    val result = DenseMatrix.zeros[Double](input1.rows, input2.rows)
    for (i <- 0 until input1.rows; j <- 0 until input2.rows) result.update(i, j, func(input1(i, ::).t, input2(j, ::).t))
    result
  }
  // the implicit helper classes of MatrixUtils.scala:135-179 (IntDividingVector, ...) are plain Scala and stay as they are
}
