package gpcore

/** JNI surface of libgpcore.so (bindings/jni/gpcore_jni.c).  Matrices cross as (data, offset, majorStride). */
object Native {
  System.loadLibrary("gpcore_jni")
  @native def ctxCreate(device: Int): Long
  @native def ctxDestroy(ctx: Long): Unit
  @native def ctxTrim(ctx: Long): Unit
  @native def dgramRbf(ctx: Long, x: Array[Double], xoff: Int, n: Int, d: Int, ldx: Int, theta: Array[Double], pos: Int, out: Array[Double]): Unit
  @native def gramRbf(ctx: Long, x: Array[Double], xoff: Int, n: Int, d: Int, ldx: Int, theta: Array[Double], out: Array[Double]): Unit
  @native def fitRbf(ctx: Long, x: Array[Double], xoff: Int, n: Int, d: Int, ldx: Int, y: Array[Double], theta: Array[Double], sigmaNoiseOrNaN: Double): Long
  @native def modelGet(ctx: Long, model: Long, what: Int, out: Array[Double], ld: Int): Unit
  @native def modelDestroy(model: Long): Unit
  @native def predict(ctx: Long, model: Long, xs: Array[Double], xsoff: Int, m: Int, ldxs: Int, mean: Array[Double], variance: Array[Double], cov: Array[Double]): Unit
  @native def epLmlRbfBatched(ctx: Long, x: Array[Double], n: Int, d: Int, ldx: Int, y: Array[Int], thetas: Array[Double], b: Int, stopEps: Double, maxSweeps: Int, strict: Boolean, lml: Array[Double], sweeps: Array[Int], info: Array[Int]): Unit
  @native def optimizeRbf(ctx: Long, x: Array[Double], n: Int, d: Int, ldx: Int, y: Array[Double], thetaInOut: Array[Double], nparams: Int, sigmaNoiseOrNaN: Double, maxIter: Int, history: Int): Double
  @native def lmlGradBatched(ctx: Long, x: Array[Double], n: Int, d: Int, ldx: Int, y: Array[Double], thetas: Array[Double], b: Int, nparams: Int, sigmaNoiseOrNaN: Double, lml: Array[Double], grad: Array[Double], info: Array[Int]): Unit
  @native def potrfLower(ctx: Long, a: Array[Double], off: Int, n: Int, lda: Int): Unit
  @native def trsmLower(ctx: Long, trans: Int, l: Array[Double], loff: Int, n: Int, ldl: Int, b: Array[Double], boff: Int, nrhs: Int, ldb: Int): Unit
  @native def epCreate(ctx: Long, k: Array[Double], off: Int, n: Int, ldk: Int, targets: Array[Int]): Long
  @native def epSweep(ctx: Long, ep: Long, nsweeps: Int, tau: Array[Double], nu: Array[Double]): Unit
  @native def epLml(ctx: Long, ep: Long, strict: Boolean): Double
  @native def epGet(ctx: Long, ep: Long, what: Int, out: Array[Double], ld: Int): Unit
  @native def epPredict(ctx: Long, ep: Long, ks: Array[Double], off: Int, m: Int, ldks: Int, kssDiag: Array[Double], prob: Array[Double]): Unit
  @native def epDestroy(ep: Long): Unit

  /** one context per JVM unless the caller builds its own; device from -Dgpcore.device (default 0) */
  lazy val defaultCtx: Long = ctxCreate(Integer.getInteger("gpcore.device", 0))
}
