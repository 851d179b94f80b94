from .loss_functions import EpochState, Loss, create_loss, load_loss_config  # noqa: F401
