"""multirate.jl_amd -- MI355X-native engine for Multirate.jl's FIRFilter / filt / filt! hot path.

This package is the host-side mirror of the reference's operator interface
(``FIRFilter``, ``filt``, ``filt!`` -> ``filt_``, ``taps2pfb``, ``outputlength``, ``inputlength``,
``reset``: export list of /root/reference/src/Multirate.jl:26-41) on top of the C ABI in
``include/multirate_hip.h`` (``libmultirate_hip.so``, hand-written gfx950 HIP kernels in ``csrc/``).

The Julia binding a Multirate.jl maintainer would use is ``julia/MultirateHIP.jl``; this Python
mirror exists because the build image has no Julia, and is what the parity tests and ``bench.py``
drive.  All arithmetic happens in the HIP library: there is no CPU fallback here, and importing
this package never touches ``oracle/``.

The directory name contains a dot, so import it through ``__graft_entry__.load_package()``
(registers it as ``multirate_jl_amd``).
"""
from __future__ import annotations

from .host import (  # noqa: F401
    ChunkRing,
    FIRFilter,
    FilterCascade,
    MultiStream,
    MultirateHIPError,
    NUMERICS_FUSED,
    NUMERICS_STRICT,
    ShardedFIRFilter,
    filt,
    filt_,
    filt_multi,
    inputlength,
    library_path,
    load_library,
    nextphase,
    outputlength,
    polyfit,
    reset,
    setphase,
    taps2pfb,
    tapsforphase,
)
from .design import (BANDPASS, BANDSTOP, HIGHPASS, LOWPASS, firdes, firprototype, kaiser,  # noqa: F401
                     kaiserlength)
from .sharding import ChannelShardedFilter, TimeShardedFilter, shard_channels, shard_time  # noqa: F401

__all__ = [
    "FIRFilter", "ChunkRing", "ShardedFIRFilter", "FilterCascade", "MultiStream", "filt", "filt_", "filt_multi", "taps2pfb", "outputlength", "inputlength", "reset", "nextphase",
    "setphase", "tapsforphase", "polyfit", "firdes", "firprototype", "kaiserlength", "kaiser", "LOWPASS", "BANDPASS", "HIGHPASS", "BANDSTOP", "ChannelShardedFilter", "TimeShardedFilter", "shard_channels", "shard_time", "load_library",
    "library_path", "MultirateHIPError", "NUMERICS_STRICT", "NUMERICS_FUSED",
]
