"""Mirror of utils/IOUtilities.scala:13-48: the two on-disk formats of the reference's fixtures.
  csvFileToDenseMatrix   separator-delimited numeric rows; rows that do not parse (headers, `?` entries in cancer.csv) are SKIPPED
  writeVectorsToFile     equal-length vectors side by side, every value followed by a TAB, one row per line
                         (co2/maunaLoa2D.txt, co2/co2PredResults*.txt, boston/bostonPredResults.txt)"""
import numpy as np


def csvFileToDenseMatrix(file, sep=","):   # :13-29
    with open(file) as fh:
        lines = fh.read().splitlines()
    parsed = [[t for t in ln.split(sep) if t != ""] for ln in lines]
    colSize = len(parsed[1])                     # seqFromReader(1): the SECOND line fixes the column count
    rows = []
    for toks in parsed:
        try:
            rows.append([float(t) for t in toks])
        except ValueError:                       # case _:Exception => collectedRows
            continue
    out = np.zeros((len(rows), colSize), order="F")
    for i, r in enumerate(rows):
        out[i, :] = r                            # a row of another length fails here, like `result(indx,::) := vector.t`
    return out


def writeVectorsToFile(file, *vecs):   # :31-48
    vecs = [np.asarray(v).reshape(-1) for v in vecs]
    if not all(v.size == vecs[0].size for v in vecs):
        raise ValueError("requirement failed: Vectors should have the same size")
    with open(file, "w") as fh:
        for i in range(vecs[0].size):
            fh.write("".join("%r\t" % (v[i].item() if hasattr(v[i], "item") else v[i]) for v in vecs) + "\n")


def readVectorsFile(file):
    """The inverse of writeVectorsToFile: columns of the tab-separated file as a matrix (rows x vectors)."""
    with open(file) as fh:
        rows = [[float(t) for t in ln.split("\t") if t.strip() != ""] for ln in fh.read().splitlines() if ln.strip() != ""]
    return np.asfortranarray(np.array(rows, dtype=np.float64))
