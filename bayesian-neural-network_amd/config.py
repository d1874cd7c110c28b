"""Drop-in `config` module: the global DEVICE plus the three constant namespaces the
reference's trainers read (reference config.py:6-7, :9-58).  On PyTorch-ROCm
``torch.cuda.is_available()`` is true on an MI355X box, so DEVICE selects the HIP device."""
import torch

DEVICE = torch.device("cuda" if torch.cuda.is_available() else "cpu")


def _constants(name, **values):
    return type(name, (), dict(values))


_INIT = dict(mu_init=[-0.2, 0.2], rho_init=[-5, -4])

RegConfig = _constants(
    "RegConfig", save_dir="./saved_models", train_size=1024, batch_size=128, lr=1e-3, epochs=1000,
    train_samples=5, test_samples=10, num_test_points=400, mode="regression", mixture_prior=False,
    hidden_units=400, noise_tolerance=.1, prior_init=[1], regression_clusters=False, **_INIT)

RLConfig = _constants(
    "RLConfig", data_dir="data/agaricus-lepiota.data", batch_size=64, num_batches=64, buffer_size=64 * 64,
    lr=1e-4, training_steps=50000, mode="regression", hidden_units=100, mixture_prior=True,
    prior_init=[0.5, -0, -6], **_INIT)

ClassConfig = _constants(
    "ClassConfig", batch_size=128, lr=1e-4, epochs=300, hidden_units=1200, mode="classification",
    train_samples=2, test_samples=10, x_shape=28 * 28, classes=10, prior_init=[1.], mixture_prior=False,
    save_dir="./saved_models", local_reparam=True, **_INIT)
