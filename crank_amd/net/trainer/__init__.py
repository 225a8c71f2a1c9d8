from .basetrainer import BaseTrainer, TrainerWrapper  # noqa: F401
from .trainer_vqvae import VQVAETrainer  # noqa: F401
from .trainer_lsgan import LSGANTrainer  # noqa: F401
from .trainer_cyclegan import CycleGANTrainer  # noqa: F401
from .trainer_stargan import StarGANTrainer  # noqa: F401
