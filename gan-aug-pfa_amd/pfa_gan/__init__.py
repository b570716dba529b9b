"""pfa_gan - MI355X-native Pix2Pix GAN step for GAN-AUG-PFA (host side; kernels live in libpfa_hip.so)."""
from .models import NLayerDiscriminator, UNetGenerator, UnetSkipConnectionBlock  # noqa: F401
from .trainer import (FusedAdam, GanTrainer, allreduce_mean_grads, broadcast_model, load_training_state,  # noqa: F401
                      train_gan_one_epoch)
from .inference import TilePipeline  # noqa: F401
from .kernels import Comm  # noqa: F401
