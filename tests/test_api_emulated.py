"""API-level parity of the drop-in classes on the CPU: the bodies of tests/api_cases.py with the C ABI served by the
numpy emulator (host logic: autograd bridges, HipAdam gather/step/state, fused trainer, fit loop, tiling).  The same
bodies run against the HIP library in tests/test_gpu_api.py."""
import pytest
import torch

import api_cases as A
from emu_backend import EmuBackend
from nirgan_hip import lib as L

torch.set_num_threads(4)
DEV = "cpu"


@pytest.fixture()
def emu():
    be = EmuBackend()
    L.set_backend(be)
    yield be
    L.set_backend(None)


@pytest.mark.parametrize("name", ["f1_g6_d.npz", "f1_g9_rs_pad.npz", "f1_inject.npz"])
def test_px2px_pl_as_lightning_drives_it(emu, golden_dir, name):
    A.px2px_pl_lightning_sequence(DEV, golden_dir, name, A.CPU_TOL)


@pytest.mark.parametrize("name", ["f1_g6_d.npz", "f1_g9_rs_pad.npz"])
def test_px2px_pl_train_batch(emu, golden_dir, name):
    A.px2px_pl_train_batch(DEV, golden_dir, name, A.CPU_TOL)


def test_pix2pix_model_optimize_parameters(emu, golden_dir):
    A.pix2pix_model_optimize_parameters(DEV, golden_dir, A.CPU_TOL)


def test_fit_loop_schedulers_checkpoint_resume(emu, tmp_path):
    A.fit_loop_schedulers_checkpoint_resume(DEV, tmp_path, A.CPU_TOL)


def test_tiled_inference_and_checkpoint_loading(emu, golden_dir, tmp_path):
    A.tiled_inference_and_checkpoint_loading(DEV, golden_dir, tmp_path, A.CPU_TOL)


def test_lightning_toggled_sequence_reuses_the_forward(emu, golden_dir):
    A.lightning_toggled_sequence_reuses_the_forward(DEV, golden_dir, A.CPU_TOL)


def test_ganloss_labels_and_adam_without_gradients(emu, golden_dir):
    A.ganloss_labels_and_adam_without_gradients(DEV, golden_dir, A.CPU_TOL)


def test_two_generator_graphs_on_one_input(emu, golden_dir):
    A.two_generator_graphs_on_one_input(DEV, golden_dir, A.CPU_TOL)


def test_px2px_pl_from_the_reference_config_key_set(emu, golden_dir):
    A.reference_config_key_set(DEV, golden_dir, A.CPU_TOL, full_width=False)
