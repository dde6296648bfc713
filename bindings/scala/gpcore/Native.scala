package gpcore

/** Thrown by the JNI glue for GP_ENOTPD; the shims rethrow it as the exception breeze.linalg.cholesky throws. */
class NotPositiveDefiniteException(msg: String) extends RuntimeException(msg)

/** JNI surface of libgpcore.so (bindings/jni/gpcore_jni.c).  Matrices cross as (data, offset, majorStride); the glue copies
  * the addressed span out of the Java array before the library runs and writes results back afterwards (no JNI critical
  * region is open across a kernel launch or a stream wait). */
object Native {
  System.loadLibrary("gpcore_jni")
  @native def ctxCreate(device: Int): Long
  @native def ctxDestroy(ctx: Long): Unit
  @native def ctxTrim(ctx: Long): Unit
  @native def gramRbf(ctx: Long, x: Array[Double], xoff: Int, n: Int, d: Int, ldx: Int, theta: Array[Double], out: Array[Double]): Unit
  @native def dgramRbf(ctx: Long, x: Array[Double], xoff: Int, n: Int, d: Int, ldx: Int, theta: Array[Double], pos: Int, out: Array[Double]): Unit
  @native def crossGramRbf(ctx: Long, xs: Array[Double], xsoff: Int, m: Int, ldxs: Int, x: Array[Double], xoff: Int, n: Int, ldx: Int, d: Int, theta: Array[Double], out: Array[Double]): Unit
  @native def fitRbf(ctx: Long, x: Array[Double], xoff: Int, n: Int, d: Int, ldx: Int, y: Array[Double], theta: Array[Double], sigmaNoiseOrNaN: Double): Long
  @native def fitFromGram(ctx: Long, k: Array[Double], koff: Int, n: Int, ldk: Int, y: Array[Double]): Long
  @native def modelGet(ctx: Long, model: Long, what: Int, out: Array[Double], ld: Int): Unit
  @native def modelDestroy(model: Long): Unit
  @native def predict(ctx: Long, model: Long, xs: Array[Double], xsoff: Int, m: Int, d: Int, ldxs: Int, mean: Array[Double], variance: Array[Double], cov: Array[Double]): Unit
  @native def posteriorFromFactor(ctx: Long, x: Array[Double], xoff: Int, n: Int, d: Int, ldx: Int, theta: Array[Double], l: Array[Double], loff: Int, ldl: Int, alpha: Array[Double], xs: Array[Double], xsoff: Int, m: Int, ldxs: Int, mean: Array[Double], variance: Array[Double], cov: Array[Double], v: Array[Double]): Unit
  @native def posteriorFromGram(ctx: Long, ks: Array[Double], m: Int, n: Int, kss: Array[Double], l: Array[Double], loff: Int, ldl: Int, alpha: Array[Double], mean: Array[Double], cov: Array[Double], v: Array[Double]): Unit
  @native def predictFromGram(ctx: Long, model: Long, ks: Array[Double], m: Int, n: Int, kss: Array[Double], mean: Array[Double], cov: Array[Double]): Unit
  @native def lmlGradBatched(ctx: Long, x: Array[Double], xoff: Int, n: Int, d: Int, ldx: Int, y: Array[Double], thetas: Array[Double], b: Int, nparams: Int, sigmaNoiseOrNaN: Double, lml: Array[Double], grad: Array[Double], info: Array[Int]): Unit
  @native def lmlGradFromGram(ctx: Long, k: Array[Double], koff: Int, n: Int, ldk: Int, y: Array[Double], dks: Array[Array[Double]], lddk: Int, sigmaNoiseOrNaN: Double, grad: Array[Double]): Double
  @native def optimizeRbf(ctx: Long, x: Array[Double], xoff: Int, n: Int, d: Int, ldx: Int, y: Array[Double], thetaInOut: Array[Double], nparams: Int, sigmaNoiseOrNaN: Double, maxIter: Int, history: Int): Double
  @native def epLmlRbfBatched(ctx: Long, x: Array[Double], xoff: Int, n: Int, d: Int, ldx: Int, y: Array[Int], thetas: Array[Double], b: Int, stopEps: Double, maxSweeps: Int, strict: Boolean, lml: Array[Double], sweeps: Array[Int], info: Array[Int]): Unit
  @native def epLmlGradRbfBatched(ctx: Long, x: Array[Double], xoff: Int, n: Int, d: Int, ldx: Int, y: Array[Int], thetas: Array[Double], b: Int, stopEps: Double, maxSweeps: Int, strict: Boolean, lml: Array[Double], grad: Array[Double], sweeps: Array[Int], info: Array[Int]): Unit
  @native def potrfLower(ctx: Long, a: Array[Double], off: Int, n: Int, lda: Int): Unit
  @native def trsmLower(ctx: Long, trans: Int, l: Array[Double], loff: Int, n: Int, ldl: Int, b: Array[Double], boff: Int, nrhs: Int, ldb: Int): Unit
  @native def invLower(ctx: Long, l: Array[Double], loff: Int, n: Int, ldl: Int, out: Array[Double]): Unit
  @native def epCreate(ctx: Long, k: Array[Double], off: Int, n: Int, ldk: Int, targets: Array[Int]): Long
  @native def epSweep(ctx: Long, ep: Long, nsweeps: Int, n: Int, tau: Array[Double], nu: Array[Double]): Unit
  @native def epSetSiteParams(ctx: Long, ep: Long, n: Int, tau: Array[Double], nu: Array[Double]): Unit
  @native def epLml(ctx: Long, ep: Long, strict: Boolean): Double
  @native def epLmlGradRbf(ctx: Long, ep: Long, x: Array[Double], xoff: Int, n: Int, d: Int, ldx: Int, theta: Array[Double], strict: Boolean, grad: Array[Double]): Unit
  @native def epGet(ctx: Long, ep: Long, what: Int, out: Array[Double], ld: Int): Unit
  @native def epPredict(ctx: Long, ep: Long, ks: Array[Double], off: Int, m: Int, n: Int, ldks: Int, kssDiag: Array[Double], prob: Array[Double]): Unit
  @native def epDestroy(ep: Long): Unit
  @native def epOptimizeRbf(ctx: Long, x: Array[Double], xoff: Int, n: Int, d: Int, ldx: Int, y: Array[Int], thetaInOut: Array[Double], stopEps: Double, maxSweeps: Int, strict: Boolean, maxIter: Int, history: Int): Double
  // batched small-n posteriors: GPOptimizer (GP-UCB), GPUnscentedKalmanFilter (GP-UKF)
  @native def smallFit(ctx: Long, x: Array[Double], xoff: Int, n: Int, d: Int, ldx: Int, y: Array[Double], g: Int, thetas: Array[Double], sigmaNoiseOrNaN: Double, capacity: Int): Long
  /** G models from factors the caller holds -- the (L, alpha) of preComputeComponents that GPUnscentedKalmanFilter.scala:116-132 keeps per state dimension; ls / alphas: g blocks of n x n (ld n) / n */
  @native def smallFromFactors(ctx: Long, x: Array[Double], xoff: Int, n: Int, d: Int, ldx: Int, thetas: Array[Double], g: Int, ls: Array[Double], alphas: Array[Double], capacity: Int): Long
  @native def smallSize(small: Long): Array[Int]      // (n, capacity, g)
  @native def smallGet(ctx: Long, small: Long, what: Int, g: Int, out: Array[Double], ld: Int): Unit
  @native def smallDestroy(small: Long): Unit
  @native def smallPosterior(ctx: Long, small: Long, g: Int, xs: Array[Double], xsoff: Int, m: Int, d: Int, ldxs: Int, mean: Array[Double], variance: Array[Double]): Unit
  @native def smallUcb(ctx: Long, small: Long, g: Int, xs: Array[Double], xsoff: Int, m: Int, d: Int, ldxs: Int, kappa: Double, value: Array[Double], grad: Array[Double]): Unit
  @native def smallAppend(ctx: Long, small: Long, d: Int, g: Int, xNew: Array[Double], yNew: Array[Double]): Unit
  @native def smallMaximizeUcb(ctx: Long, small: Long, g: Int, starts: Array[Double], c: Int, d: Int, kappa: Double, maxIter: Int, history: Int, bestX: Array[Double]): Double
  // Co2Kernel
  @native def gramCo2(ctx: Long, x: Array[Double], n: Int, theta: Array[Double], out: Array[Double]): Unit
  @native def dgramCo2(ctx: Long, x: Array[Double], n: Int, theta: Array[Double], pos: Int, out: Array[Double]): Unit
  @native def crossGramCo2(ctx: Long, xs: Array[Double], m: Int, x: Array[Double], n: Int, theta: Array[Double], out: Array[Double]): Unit
  @native def fitCo2(ctx: Long, x: Array[Double], n: Int, y: Array[Double], theta: Array[Double], sigmaNoiseOrNaN: Double): Long
  @native def lmlGradCo2Batched(ctx: Long, x: Array[Double], n: Int, y: Array[Double], thetas: Array[Double], b: Int, nparams: Int, sigmaNoiseOrNaN: Double, lml: Array[Double], grad: Array[Double], info: Array[Int]): Unit
  @native def optimizeCo2(ctx: Long, x: Array[Double], n: Int, y: Array[Double], thetaInOut: Array[Double], nparams: Int, sigmaNoiseOrNaN: Double, maxIter: Int, history: Int): Double
  // multi-GPU: one JVM per GPU, RCCL all-gather of the per-rank results
  @native def distUniqueId(ctx: Long): Array[Byte]
  @native def distInit(ctx: Long, id: Array[Byte], rank: Int, world: Int): Long
  @native def distDestroy(dist: Long): Unit
  @native def distLmlGradBatched(ctx: Long, dist: Long, x: Array[Double], xoff: Int, n: Int, d: Int, ldx: Int, y: Array[Double], thetas: Array[Double], b: Int, nparams: Int, sigmaNoiseOrNaN: Double, lml: Array[Double], grad: Array[Double], info: Array[Int]): Unit
  @native def distPredict(ctx: Long, dist: Long, model: Long, xs: Array[Double], xsoff: Int, m: Int, d: Int, ldxs: Int, mean: Array[Double], variance: Array[Double]): Unit

  /** one context per JVM unless the caller builds its own; device from -Dgpcore.device (default 0) */
  lazy val defaultCtx: Long = ctxCreate(Integer.getInteger("gpcore.device", 0))

  /** give cached device workspaces back (gp_ctx_trim) / release the default context at JVM shutdown */
  def trim(): Unit = ctxTrim(defaultCtx)
  def shutdown(): Unit = ctxDestroy(defaultCtx)

  /** -Dgpcore.strict=false selects the intended EP formulas instead of the reference's as-compiled ones (SURVEY.md A19/A22) */
  lazy val strict: Boolean = java.lang.Boolean.parseBoolean(System.getProperty("gpcore.strict", "true"))

  /** a column-major, non-transposed view the C-ABI can address as (data, offset, majorStride); anything else is copied */
  def dense(m: breeze.linalg.DenseMatrix[Double]): breeze.linalg.DenseMatrix[Double] =
    if (m.isTranspose || m.majorStride < m.rows) m.copy else m

  /** breeze.linalg.cholesky's failure (breeze 0.8.1: `throw new NotConvergedException(NotConvergedException.Iterations)`) */
  def rethrowNotPd[T](body: => T): T =
    try body catch {
      case e: NotPositiveDefiniteException =>
        throw new breeze.linalg.NotConvergedException(breeze.linalg.NotConvergedException.Iterations, e.getMessage)
    }
}
