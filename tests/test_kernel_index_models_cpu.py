"""The numpy models of the small-linear kernels' index arithmetic (bounds-checked, write-once) on a few small shapes."""
import pytest

from index_models import small_linear_bwd16, small_linear_bwd32, small_linear_fwd


@pytest.mark.parametrize("R,inn,out", [(37, 50, 33), (32, 64, 1), (70, 24, 40)])
def test_backward_index_models(R, inn, out):
    small_linear_bwd32.run(R, inn, out)
    small_linear_bwd16.run(R, inn, out)
    small_linear_bwd16.run(R, inn, out, want_dx=False, want_db=False)


@pytest.mark.parametrize("R,inn,out", [(37, 50, 33), (20, 260, 24), (1, 1, 1)])
def test_forward_index_model(R, inn, out):
    small_linear_fwd.run(R, inn, out, relu=True)
