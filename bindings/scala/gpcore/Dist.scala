package gpcore

import breeze.linalg.{DenseMatrix, DenseVector}

/** One JVM per GPU (SURVEY.md 8(e)): this rank's RCCL communicator on the default context's device and the two sharded calls.
  * The path shards over INDEPENDENT units only -- hyper-parameter settings (GpPredictor.logLikelihoodWithDerivatives evaluated by
  * obtainOptimalHyperParams / a mesh, gp/regression/GpPredictor.scala:60-80,126-142) and test points (GpPredictor.predict, :24-43):
  * every rank passes the SAME arguments, evaluates the contiguous slice [rank * ceil(U / world), ...) on its own GPU and receives
  * the assembled result; the only collective is one all-gather of the results (after one of a status word per rank, so a rank
  * that fails locally cannot strand the others: they get `PeerFailedException`).
  *
  * `exchange` ships rank 0's 128-byte id to every rank over whatever channel the hosts already have (a file, a socket, the
  * cluster manager's broadcast): it is called with Some(id) on rank 0 and None elsewhere and returns the id on every rank. */
class DistGroup(val rank: Int, val world: Int, exchange: Option[Array[Byte]] => Array[Byte]) {
  import Native.{defaultCtx => ctx, dense}

  private val handle: Long = {
    val id = exchange(if (rank == 0) Some(Native.distUniqueId(ctx)) else None)
    Native.distInit(ctx, id, rank, world)
  }

  /** (LML, gradient) of every row of `thetas` (B x (d+2)); setting b is evaluated by rank b / ceil(B / world).
    * info(b) > 0: the setting's K was not positive definite (1-based failing pivot), its LML is NaN. */
  def lmlGradBatched(x: DenseMatrix[Double], y: DenseVector[Double], thetas: DenseMatrix[Double], sigmaNoise: Option[Double] = None):
      (DenseVector[Double], DenseMatrix[Double], Array[Int]) = {
    val xc = dense(x); val (b, p) = (thetas.rows, thetas.cols)
    require(p == xc.cols + 2 && y.length == xc.rows)
    val lml = new Array[Double](b); val grad = new Array[Double](b * p); val info = new Array[Int](b)
    Native.distLmlGradBatched(ctx, handle, xc.data, xc.offset, xc.rows, xc.cols, xc.majorStride, y.toArray, thetas.t.copy.data, b, p,
      sigmaNoise.getOrElse(Double.NaN), lml, grad, info)
    (DenseVector(lml), new DenseMatrix(p, b, grad).t, info)                    // grad is B x P row-major
  }

  /** posterior mean and variance of the m rows of `xs` against a fitted model (a handle of Native.fitRbf): test point i is
    * evaluated by rank i / ceil(m / world) */
  def predict(model: Long, xs: DenseMatrix[Double]): (DenseVector[Double], DenseVector[Double]) = {
    val t = dense(xs)
    val mean = new Array[Double](t.rows); val variance = new Array[Double](t.rows)
    Native.distPredict(ctx, handle, model, t.data, t.offset, t.rows, t.cols, t.majorStride, mean, variance)
    (DenseVector(mean), DenseVector(variance))
  }

  def close() { Native.distDestroy(handle) }
}
