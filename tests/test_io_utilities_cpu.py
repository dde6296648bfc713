"""The reference's on-disk formats (utils/IOUtilities.scala:13-48, gp/regression/Co2Prediction.scala:139-183) against its own data
files kept as fixtures under tests/golden/ (boston.csv, cancer.csv, co2/maunaLoa.txt, the head of co2/maunaLoa2D.txt)."""
import os

import numpy as np

from gp_algos_amd.gp.regression.co2_prediction import co2DataToYearWithValue, loadInput
from gp_algos_amd.utils.io_utilities import csvFileToDenseMatrix, readVectorsFile, writeVectorsToFile

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_csv_loader_on_the_reference_data_sets():
    boston = csvFileToDenseMatrix(os.path.join(GOLD, "boston.csv"), sep=" ")
    assert boston.shape == (506, 14) and boston[0, -1] == 24.0 and boston.flags.f_contiguous
    cancer = csvFileToDenseMatrix(os.path.join(GOLD, "cancer.csv"))
    assert cancer.shape[1] == 11 and cancer.shape[0] == 683           # the 16 rows holding `?` are skipped (IOUtilities.scala:19-25)
    assert set(np.unique(cancer[:, -1])) == {2.0, 4.0}


def test_mauna_loa_loader_and_year_value_conversion_match_the_stored_2d_file():
    m = loadInput(os.path.join(GOLD, "co2", "maunaLoa.txt"))
    assert m.shape[1] == 14 and m[0, 0] == 1958.0 and m[0, 1] == -99.99
    whole, rest = co2DataToYearWithValue(m, 1.0)                       # MasterThesisRelatedTasks.writeCo2DataSetToFile
    assert whole.shape == (607, 2) and rest.shape == (0, 2)
    ref = readVectorsFile(os.path.join(GOLD, "co2", "maunaLoa2D_head.txt"))
    assert np.array_equal(whole[:ref.shape[0]], ref)                   # year + (month - 1)/12 and ppm, bit for bit
    train, test = co2DataToYearWithValue(m, 0.7)
    assert train.shape == (424, 2) and test.shape == (183, 2)


def test_vector_file_round_trip(tmp_path):
    a, b = np.array([1958.1666666666667, 1958.25, 2.0]), np.array([315.7, 317.45, -0.5])
    f = tmp_path / "v.txt"
    writeVectorsToFile(f, a, b)
    text = f.read_text()
    assert text.splitlines()[0] == "1958.1666666666667\t315.7\t"       # every value is followed by a TAB, like the Scala writer
    back = readVectorsFile(f)
    assert np.array_equal(back[:, 0], a) and np.array_equal(back[:, 1], b)
