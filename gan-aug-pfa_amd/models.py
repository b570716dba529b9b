"""Drop-in for the reference's models.py: `from models import UNetGenerator, NLayerDiscriminator` (train_gan.py:14,
generate_synthetic_data.py:10) and `from models import SiameseUNet` (train.py:16, evaluate.py:12) keep working when this
directory is first on sys.path.  The GAN classes run on the HIP engine; SiameseUNet is plain PyTorch (out of the
accelerated path)."""
from pfa_gan.models import NLayerDiscriminator, UNetGenerator, UnetSkipConnectionBlock  # noqa: F401
from pfa_gan.siamese import AttentionGate, SiameseUNet, double_conv  # noqa: F401
