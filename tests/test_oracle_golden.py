"""The oracle restatement vs the golden vectors produced by the reference itself
(tests/golden/make_golden.py), and - when /root/reference is present - vs the live
reference.  CPU only."""
import importlib.util
import os

import numpy as np
import pytest
import torch

from cotr_amd.utils.synth import state_checksum
from oracle import cotr_oracle, ref_import

_spec = importlib.util.spec_from_file_location(
    'make_golden', os.path.join(os.path.dirname(__file__), 'golden', 'make_golden.py'))
make_golden = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(make_golden)

CASES = list(make_golden.CASES)
PX_BAR = 1e-3  # BASELINE.json north_star: 1e-3 px, fp32


@pytest.mark.parametrize('name', CASES)
def test_oracle_matches_reference_golden(name, golden_dir):
    g = np.load(os.path.join(golden_dir, name + '.npz'))
    sd, img, qs = make_golden.case_inputs(name)
    # the seeded generator reproduces the weights the golden file was made with
    assert np.allclose(state_checksum(sd), g['weights_checksum'], rtol=1e-12)
    taps = {}
    o32 = cotr_oracle.cotr_forward(sd, img, qs, taps=taps)
    assert o32.shape == g['pred_f32'].shape
    # fp32 restatement vs fp32 reference and vs the fp64 reference: inside the bar.  The ill-conditioned softmax-extreme
    # cases (peaky16 / peaky32: near one-hot attention in all 12 layers) put the reference's OWN fp32 run 1.7e-2 / 3.2 px
    # from its fp64 run; there the restatement is held to 3x that gap from the fp64 truth, and the fp64 restatement to
    # the fp64 reference exactly (test below)
    ref_gap = cotr_oracle.px_err(torch.from_numpy(g['pred_f32']), torch.from_numpy(g['pred_f64']))
    bar = max(PX_BAR, 3 * ref_gap)
    assert cotr_oracle.px_err(o32, torch.from_numpy(g['pred_f64'])) < bar
    if ref_gap < PX_BAR / 3:
        assert cotr_oracle.px_err(o32, torch.from_numpy(g['pred_f32'])) < PX_BAR
        mem = taps['enc.5'].permute(1, 0, 2)[:, ::8]            # [B,64,256]
        assert float((mem - torch.from_numpy(g['memory'])).abs().max()) < 5e-5


@pytest.mark.parametrize('name', ['single_b1_q1', 'peaky_b1_q64', 'peaky16_b1_q64', 'peaky32_b1_q64', 'flat_b1_q64'])
def test_oracle_fp64_is_the_reference_fp64(name, golden_dir):
    g = np.load(os.path.join(golden_dir, name + '.npz'))
    sd, img, qs = make_golden.case_inputs(name)
    o64 = cotr_oracle.cotr_forward(sd, img, qs, dtype=torch.float64)
    assert cotr_oracle.px_err(o64, torch.from_numpy(g['pred_f64'])) < 1e-5


@pytest.mark.skipif(not ref_import.reference_available(), reason='/root/reference not on this machine')
def test_restatement_matches_live_reference():
    torch.manual_seed(0)
    model = ref_import.build_reference_model()
    sd, img, qs = make_golden.case_inputs('ragged_b2_q257')
    model.load_state_dict(sd)
    with torch.no_grad():
        ref = model(img, qs)['pred_corrs']
    assert cotr_oracle.px_err(cotr_oracle.cotr_forward(sd, img, qs), ref) < PX_BAR


def test_query_independence_of_pairs():
    """Pairs never interact: a batch equals its per-pair runs."""
    sd, img, qs = make_golden.case_inputs('ragged_b2_q257')
    both = cotr_oracle.cotr_forward(sd, img, qs[:, :16])
    one = cotr_oracle.cotr_forward(sd, img[1:2], qs[1:2, :16])
    assert cotr_oracle.px_err(both[1:2], one) < 2e-4
