"""Randomised self-consistency runs (tools/debug/*_fuzz.py) of the three proof-carrying shortcuts of the hot path, each
against the exhaustive / row-evaluating form of the same computation on the device: feature-space and 3-D kNN filters
(csrc/knn_filter.hip, knn_normal.hip) vs the exact kernels of csrc/knn.hip; the filtered segment diameter
(csrc/segdiam.hip) vs softgroup.hip:seg_diameter_kernel; ball query with thresholds <= 0 (softgroup.hip) vs
GCANET_BQ_EXACT=1; and the bf16 / fp16 flash attention kernels (attention_mfma.hip) vs the exact f32 kernel within the
16-bit budget; the bf16 / IEEE-half EdgeConv kernels vs the exact f32 kernel on representable operands at 1e-4.
The tools take (seed, cases); the developer runs used seeds 1-5 with 150-200 cases each."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("tool,seed,cases", [("knn_fuzz.py", 11, 30), ("segdiam_fuzz.py", 11, 40), ("bq_fuzz.py", 11, 12),
                                             ("attn_fuzz.py", 11, 40), ("edgeconv_fuzz.py", 11, 25)])
def test_fuzz_tool_reports_no_mismatch(dev, tool, seed, cases):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "debug", tool), str(seed), str(cases)],
                         capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    assert "mismatches 0" in out.stdout, out.stdout[-2000:]
