"""In-memory model the hot path reads and writes -- the host-side mirror of
include/ReadExperiment.hpp:65-99,236-257, include/Transcript.hpp:14-16,49-81,204-206 and
include/SailfishOpts.hpp:9-41, laid out for the device: the reference's AoS
`std::vector<Transcript>` becomes one SoA block resident in HBM."""
from dataclasses import dataclass, field
from typing import List, Optional

import numpy as np
import torch


@dataclass
class SailfishOpts:
    """Fields (and defaults) of include/SailfishOpts.hpp that the hot path consults;
    defaults as set by the CLI table src/SailfishQuantify.cpp:1079-1153."""
    numThreads: int = 1
    useVBOpt: bool = False
    noEffectiveLengthCorrection: bool = False
    useUnsmoothedFLD: bool = False
    maxFragLen: int = 1000
    numFragSamples: int = 10000
    fragLenDistPriorMean: int = 200
    fragLenDistPriorSD: int = 80
    maxReadOccs: int = 200
    numBootstraps: int = 0
    numGibbsSamples: int = 0
    biasCorrect: bool = False
    gcBiasCorrect: bool = False
    gcSampFactor: int = 1               # --gcSizeSamp
    pdfSampFactor: int = 1              # --gcSpeedSamp
    dumpEq: bool = False
    auxDir: str = "aux"
    jointLog: Optional[object] = None   # callable(level:int, msg:str)


class Transcripts:
    """SoA view of std::vector<Transcript>: RefName / RefLength / EffectiveLength host-visible,
    EffectiveLength, estCount and mass device-resident (float64)."""

    def __init__(self, names: List[str], ref_len, device="cuda"):
        ref_len = np.ascontiguousarray(ref_len, dtype=np.uint32)
        assert len(names) == len(ref_len)
        self.RefName = list(names)
        self.device = torch.device(device)
        # stored as raw uint32 bits in an int32 tensor (torch has no uint32 arithmetic)
        self.RefLength = torch.from_numpy(ref_len.view(np.int32).copy()).to(self.device)
        M = len(ref_len)
        # Transcript ctor: EffectiveLength(-1.0) (Transcript.hpp:16); set by the FLD stage
        self.EffectiveLength = torch.full((M,), -1.0, dtype=torch.float64, device=self.device)
        self.estCount = torch.zeros(M, dtype=torch.float64, device=self.device)
        self.mass = torch.zeros(M, dtype=torch.float64, device=self.device)
        self.active = None

    def __len__(self):
        return len(self.RefName)

    def ref_length_f64(self):
        return (self.RefLength.to(torch.int64) & 0xFFFFFFFF).to(torch.float64)


class ReadExperiment:
    """Owner of the transcripts, the equivalence-class builder and the fragment counters
    (include/ReadExperiment.hpp:65-99, 236-257)."""

    def __init__(self, transcripts: Transcripts, sopt: Optional[SailfishOpts] = None):
        self._transcripts = transcripts
        self._eq = None
        self._logger = sopt.jointLog if sopt else None
        self._num_mapped = 0
        self._num_observed = 0
        self._fld = None

    def transcripts(self) -> Transcripts:
        return self._transcripts

    def equivalenceClassBuilder(self):
        if self._eq is None:
            from .eqclass import EquivalenceClassBuilder
            self._eq = EquivalenceClassBuilder(logger=self._logger, device=self._transcripts.device)
        return self._eq

    def numMappedFragments(self) -> int:
        return self._num_mapped

    def setNumMappedFragments(self, n: int):
        """numMappedFragmentsAtomic() = n (ReadExperiment.hpp:73)."""
        self._num_mapped = int(n)

    def numObservedFragments(self) -> int:
        return self._num_observed

    def setNumObservedFragments(self, n: int):
        self._num_observed = int(n)

    def fragLengthDist(self):
        return self._fld

    # ---- what bias correction reads (include/ReadExperiment.hpp:93-97, 178-212, 240-255) ----
    def setSequences(self, seq, offsets):
        """The index's concatenated transcript sequence (RapMapSAIndex::seq; bytes / uint8 array / tensor) and
        txpOffsets -- what loadTranscriptsFromQuasiIndex hands to Transcript::setSequence (:108-117)."""
        dev = self._transcripts.device
        if isinstance(seq, (bytes, bytearray)):
            seq = np.frombuffer(bytes(seq), dtype=np.uint8).copy()
        if isinstance(seq, np.ndarray):
            seq = torch.from_numpy(np.ascontiguousarray(seq, dtype=np.uint8))
        self._seq = seq.to(dev)
        self._seq_off = torch.as_tensor(np.asarray(offsets, dtype=np.int64)).to(dev)

    def readBias(self):
        """ReadKmerDist<6>::counts: 4096 uint32, pseudo-count 1 (include/ReadKmerDist.hpp:17-22)."""
        if getattr(self, "_read_bias", None) is None:
            self._read_bias = np.ones(4096, np.uint32)
        return self._read_bias

    def observedGC(self):
        """101 uint32, pseudo-count 1 (include/ReadExperiment.hpp:47-51)."""
        if getattr(self, "_observed_gc", None) is None:
            self._observed_gc = np.ones(101, np.uint32)
        return self._observed_gc

    def addNumFwd(self, n): self._num_fwd = getattr(self, "_num_fwd", 0) + int(n)
    def addNumRC(self, n): self._num_rc = getattr(self, "_num_rc", 0) + int(n)
    def numFwd(self): return getattr(self, "_num_fwd", 0)
    def numRC(self): return getattr(self, "_num_rc", 0)

    def expectedSeqBias(self):
        return getattr(self, "_expected_seq", None) if getattr(self, "_expected_seq", None) is not None else np.ones(4096)

    def setExpectedSeqBias(self, v): self._expected_seq = np.asarray(v, dtype=np.float64)

    def expectedGCBias(self):
        return getattr(self, "_expected_gc", None) if getattr(self, "_expected_gc", None) is not None else np.ones(101)

    def setExpectedGCBias(self, v): self._expected_gc = np.asarray(v, dtype=np.float64)

    def biasModel(self, sopt):
        """Device handle over everything updateEffectiveLengths reads (src/SailfishUtils.cpp:611-690)."""
        from .bias import BiasModel
        if getattr(self, "_seq", None) is None:
            raise RuntimeError("bias correction needs the transcript sequences: call setSequences first")
        if self._fld is None:
            raise RuntimeError("bias correction needs the fragment length distribution: call setFragLengthDist first")
        t = self._transcripts
        return BiasModel(self._seq, self._seq_off, t.RefLength, t.EffectiveLength, self._fld.astype(np.uint32),
                         self.readBias(), self.observedGC(), self.numFwd(), self.numRC(),
                         seq_bias=sopt.biasCorrect, gc_bias=sopt.gcBiasCorrect,
                         gc_speed_samp=sopt.pdfSampFactor, gc_size_samp=sopt.gcSampFactor)

    def setFragLengthDist(self, fld):
        self._fld = np.asarray(fld, dtype=np.int32)
