"""``model_map``: name -> class, as used by every entry script (reference ``models/mapping.py:7-22``)."""
from .ablation_models import (DualModel, SequenceFpModel, SequenceModel, StructureModel, StructureModel_SSL,
                              StructureModelv2)
from .comparative_models import (HybridModel_Comparative, HybridModel_Comparative_SSL, HybridModelv2_Comparative,
                                 HybridModelv2_Comparative_SSL)
from .hybrid_models import HybridModel, HybridModel_SSL, HybridModelv2, HybridModelv2_SSL

model_map = {cls.__name__: cls for cls in (
    SequenceModel, SequenceFpModel, StructureModel, StructureModel_SSL, StructureModelv2,
    HybridModel, HybridModel_SSL, HybridModelv2, HybridModelv2_SSL,
    HybridModel_Comparative, HybridModel_Comparative_SSL, HybridModelv2_Comparative,
    HybridModelv2_Comparative_SSL, DualModel)}
