"""MODEL / BACKBONE registries - same surface as the reference's model/registry.py:1-4,
so `MODEL.get(config.model.name)(config.model)` (train.py:158-169, test.py:65-76)
builds the MI355X-native heads unchanged."""
from ..utils.repository import Repository

MODEL = Repository()
BACKBONE = Repository()


def install_into(reference_model_registry):
    """Drop-in: overwrite the reference's own entries (BCNN, CBCNN, MPN, APCNN, OSMENet)
    in ITS `model.registry.MODEL` with the classes registered here.  See INTEGRATION.md."""
    for name, factory in MODEL.items():
        if name in ('ResNet50', 'ResNet101') and name in reference_model_registry:
            continue                     # plain classifiers, no pooling head: the reference keeps its own
        dict.__setitem__(reference_model_registry, name, factory)
    return reference_model_registry
