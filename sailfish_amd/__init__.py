"""sailfish_amd -- MI355X-native quantification core for Sailfish (`sailfish quant` hot path):
equivalence-class construction and the collapsed EM/VBEM optimizer as HIP kernels behind a C ABI
(include/sfgpu.h, sailfish_amd/csrc/libsfgpu.so).  This package is the host-side mirror of the
reference's interface for that path; it holds no compute and no CPU fallback."""
from .experiment import ReadExperiment, SailfishOpts, Transcripts  # noqa: F401
from .eqclass import EquivalenceClassBuilder, EqVec, xxh64_labels  # noqa: F401
from .optimizer import CollapsedEMOptimizer, EMProblem  # noqa: F401
from .gibbs import CollapsedGibbsSampler, gibbs_sample  # noqa: F401
from . import bias, efflen, genes, hits, mapper, quant, writer  # noqa: F401
