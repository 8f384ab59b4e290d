from .sg_trainer import Accuracy, DDPNotSetupException, Top5, Trainer  # noqa: F401
