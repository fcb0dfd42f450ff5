"""The bitwise gates of the label assignment (tests/test_kmeans2.py) once more under the `gpu`
marker: the C k-means borrows the BLAS entry points of the numpy / scipy of the machine it runs
on (kmeans2.py), so the GPU box's own OpenBLAS must be the one that is tested there -- the
first-use self-test alone is not the whole gate.  Same functions, nothing GPU-specific in them."""

import pytest
from test_kmeans2 import (  # noqa: F401 - collected here under the gpu marker
    test_degenerate_inputs_match_k_means,
    test_fast_path_is_active_with_the_pinned_scikit_learn,
    test_labels_and_stream_match_k_means,
    test_native_lloyd_hands_an_empty_cluster_back,
    test_native_lloyd_iteration_equals_the_cython_kernel_start_by_start,
    test_native_seeding_equals_the_numpy_seeding_bit_for_bit,
    test_native_seeding_hands_an_emptied_cluster_back,
    test_single_thread_lloyd_equals_the_public_call_above_one_chunk,
)

pytestmark = pytest.mark.gpu
