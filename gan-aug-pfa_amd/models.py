"""Drop-in for the GAN half of the reference's models.py: `from models import UNetGenerator, NLayerDiscriminator`
(train_gan.py:14, generate_synthetic_data.py:10) keeps working when this directory is first on sys.path."""
from pfa_gan.models import NLayerDiscriminator, UNetGenerator, UnetSkipConnectionBlock  # noqa: F401
