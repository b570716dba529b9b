"""pfa_gan - MI355X-native Pix2Pix GAN step for GAN-AUG-PFA (host side; kernels live in libpfa_hip.so)."""
from .models import NLayerDiscriminator, UNetGenerator, UnetSkipConnectionBlock  # noqa: F401
from .trainer import FusedAdam, GanTrainer, train_gan_one_epoch  # noqa: F401
