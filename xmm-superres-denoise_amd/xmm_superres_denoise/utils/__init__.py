from .loss_functions import Loss, create_loss, load_loss_config  # noqa: F401
