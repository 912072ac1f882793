"""include/idto/optimizer/penta_diagonal_matrix.h against the reference's own tests of the class
(optimizer/test/penta_diagonal_solver_test.cc:42-107: SymmetricMatrixEmpty, MutateMatrix,
SymmetricMatrix - the is_symmetric flag and the throw rules of optimizer/penta_diagonal_matrix.h:131-176)
plus MultiplyBy / ExtractDiagonal / ScaleByDiagonal / MakeDense against a dense matrix.  The header is
host-only C++: the test program tests/cpp/penta_diagonal_matrix_test.cc is compiled with g++ and run."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_penta_diagonal_matrix_semantics(tmp_path):
    exe = str(tmp_path / "penta_diagonal_matrix_test")
    subprocess.run(["g++", "-O1", "-std=c++17", "-Wall", "-Werror", "-I" + os.path.join(ROOT, "include"),
                    os.path.join(ROOT, "tests", "cpp", "penta_diagonal_matrix_test.cc"), "-o", exe], check=True)
    out = subprocess.run([exe], check=True, capture_output=True, text=True).stdout
    assert "all passed" in out, out
