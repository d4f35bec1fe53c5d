"""API-level parity of the drop-in classes ON THE MI355X: what the reference's
train.py / create_synthetic_dataset.py call -- Px2Px_PL.training_step / configure_optimizers / predict_step / forward /
train_batch, Pix2PixModel.optimize_parameters, the fit loop, tiled inference and checkpoint loading -- through the
autograd bridges, HipAdam and the HIP library, against the reference's golden vectors (losses, gradients and the
parameters after the reference's torch.optim.Adam steps).  Bodies in tests/api_cases.py (shared with the CPU suite)."""
import pytest

import api_cases as A

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.mark.parametrize("name", ["f1_g6_d.npz", "f1_g9_rs_pad.npz", "f1_inject.npz"])
def test_px2px_pl_as_lightning_drives_it(golden_dir, name):
    A.px2px_pl_lightning_sequence(DEV, golden_dir, name, A.GPU_TOL)


@pytest.mark.parametrize("name", ["f1_g6_d.npz", "f1_g9_rs_pad.npz"])
def test_px2px_pl_train_batch(golden_dir, name):
    A.px2px_pl_train_batch(DEV, golden_dir, name, A.GPU_TOL)


def test_pix2pix_model_optimize_parameters(golden_dir):
    A.pix2pix_model_optimize_parameters(DEV, golden_dir, A.GPU_TOL)


def test_fit_loop_schedulers_checkpoint_resume(tmp_path):
    A.fit_loop_schedulers_checkpoint_resume(DEV, tmp_path, A.GPU_TOL)


def test_tiled_inference_and_checkpoint_loading(golden_dir, tmp_path):
    A.tiled_inference_and_checkpoint_loading(DEV, golden_dir, tmp_path, A.GPU_TOL)


def test_lightning_toggled_sequence_reuses_the_forward(golden_dir):
    A.lightning_toggled_sequence_reuses_the_forward(DEV, golden_dir, A.GPU_TOL)


def test_ganloss_labels_and_adam_without_gradients(golden_dir):
    A.ganloss_labels_and_adam_without_gradients(DEV, golden_dir, A.GPU_TOL)


def test_two_generator_graphs_on_one_input(golden_dir):
    A.two_generator_graphs_on_one_input(DEV, golden_dir, A.GPU_TOL)


def test_px2px_pl_from_the_reference_config_key_set(golden_dir):
    A.reference_config_key_set(DEV, golden_dir, A.GPU_TOL, full_width=True)
