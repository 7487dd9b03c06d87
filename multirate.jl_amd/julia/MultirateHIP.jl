# MultirateHIP.jl -- thin Julia binding of libmultirate_hip.so (include/multirate_hip.h).
#
# Drop-in for the hot path of Multirate.jl's src/Filters.jl: same type and function names
# (FIRFilter, FIRStandard/FIRDecimator/FIRInterpolator/FIRRational/FIRArbitrary/FIRFarrow, filt, filt!,
# taps2pfb, outputlength, inputlength, reset, nextphase, setphase, tapsforphase, polyfit, firdes, firprototype, kaiserlength), same argument meaning, same return values,
# same errors -- every method body is a ccall.  Modern Julia (>= 1.6) syntax; the reference is
# Julia-0.3 source and cannot be loaded by a current Julia, so this module stands beside it rather
# than patching it.  See INTEGRATION.md for how a maintainer wires it into Multirate.jl.
#
# NOTE: the build image has no Julia, so this file has not been executed there; it is kept
# deliberately free of logic (all logic lives behind the C ABI, which the Python mirror in
# ../host.py exercises symbol by symbol in tests/).
module MultirateHIP

export FIRFilter, FIRKernel, FIRStandard, FIRDecimator, FIRInterpolator, FIRRational, FIRArbitrary, FIRFarrow,
       filt, filt!, taps2pfb, outputlength, inputlength, reset, nextphase, setphase, tapsforphase, tapsforphase!, polyfit,
       firdes, firprototype, kaiserlength, kaiser, FIRResponse, LOWPASS, BANDPASS, HIGHPASS, BANDSTOP,
       FilterCascade, filt_device!, filt_device_chunked!, scheduleinfo, advancestate!, ChunkRing, pushchunks!, drain, ringinfo, ShardedFIRFilter

const libmr = get(ENV, "MRHIP_LIB_PATH", joinpath(@__DIR__, "..", "libmultirate_hip.so"))

# ---- enums of multirate_hip.h ------------------------------------------------------------------
const MRHIP_F32, MRHIP_F64, MRHIP_C64, MRHIP_C128 = Cint(0), Cint(1), Cint(2), Cint(3)
dtypecode(::Type{Float32}) = MRHIP_F32
dtypecode(::Type{Float64}) = MRHIP_F64
dtypecode(::Type{ComplexF32}) = MRHIP_C64
dtypecode(::Type{ComplexF64}) = MRHIP_C128

# kernel kinds == the reference's FIRKernel subtypes (src/Filters.jl:15-117), PARAMETRIC in the tap type like theirs
# (`type FIRRational{T} <: FIRKernel`, :62): user code that names FIRFilter{FIRRational{Float32}} loads and dispatches as it does
# on the reference.  Marker types: the fields live behind the C ABI (see KernelProxy below).
abstract type FIRKernel end
struct FIRStandard{T} <: FIRKernel end
struct FIRDecimator{T} <: FIRKernel end
struct FIRInterpolator{T} <: FIRKernel end
struct FIRRational{T} <: FIRKernel end
struct FIRArbitrary{T} <: FIRKernel end
struct FIRFarrow{T} <: FIRKernel end
const KINDS = (FIRStandard, FIRDecimator, FIRInterpolator, FIRRational, FIRArbitrary, FIRFarrow)

struct MRHIPState                      # mirror of `mrhip_state`
    kind::Int32; tap_dtype::Int32; sample_dtype::Int32; output_dtype::Int32
    nchannels::Int64; hLen::Int64; interpolation::Int64; decimation::Int64; Nphi::Int64
    tapsPerPhi::Int64; historyLen::Int64; phiIdx::Int64; inputDeficit::Int64; xIdx::Int64
    rate::Float64; phiAccumulator::Float64; alpha::Float64; delta::Float64
end

lasterror() = unsafe_string(ccall((:mrhip_last_error, libmr), Cstring, ()))
check(rc::Integer) = rc == 0 ? nothing : error(lasterror())   # reference: error("...") in the same places

# ---- FIRFilter -----------------------------------------------------------------------------------
# Like the reference's FIRFilter (src/Filters.jl:151-198), whose `history` takes the element type of
# the first x it sees (:452), the device object is created on the first filt call.
mutable struct FIRFilter{Tk<:FIRKernel}
    h::Vector
    ratio::Union{Rational{Int},Nothing}
    rate::Float64
    Nphi::Int
    polyorder::Int                    # FIRFarrow only (-1 otherwise)
    device::Int
    handle::Ptr{Cvoid}
    Tx::Union{DataType,Nothing}
    nchannels::Int
end

function kindof(ratio::Rational)
    L, M = numerator(ratio), denominator(ratio)
    ratio == 1 ? FIRStandard : L == 1 ? FIRDecimator : M == 1 ? FIRInterpolator : FIRRational   # :163-175
end

# FIRFilter(h, resampleRatio::Rational = 1//1)            src/Filters.jl:158-180
function FIRFilter(h::Vector{Th}, ratio::Rational = 1//1; device::Integer = 0) where {Th<:Union{Float32,Float64}}
    r = Rational{Int}(ratio)
    f = FIRFilter{kindof(r){Th}}(copy(h), r, 0.0, 0, -1, device, C_NULL, nothing, 0)
    finalizer(destroy!, f)
end
# FIRFilter(h, rate::AbstractFloat, Nphi = 32)            src/Filters.jl:183-189
function FIRFilter(h::Vector{Th}, rate::AbstractFloat, Nphi::Integer = 32; device::Integer = 0) where {Th<:Union{Float32,Float64}}
    rate > 0.0 || error("rate must be greater than 0")
    f = FIRFilter{FIRArbitrary{Th}}(copy(h), nothing, Float64(rate), Nphi, -1, device, C_NULL, nothing, 0)
    finalizer(destroy!, f)
end
# FIRFilter(h, rate::AbstractFloat, Nphi, polyorder)      src/Filters.jl:192-198  (FIRFarrow)
function FIRFilter(h::Vector{Th}, rate::AbstractFloat, Nphi::Integer, polyorder::Integer; device::Integer = 0) where {Th<:Union{Float32,Float64}}
    rate > 0.0 || error("rate must be greater than 0")
    f = FIRFilter{FIRFarrow{Th}}(copy(h), nothing, Float64(rate), Nphi, polyorder, device, C_NULL, nothing, 0)
    finalizer(destroy!, f)
end

function destroy!(f::FIRFilter)
    f.handle == C_NULL || ccall((:mrhip_destroy, libmr), Cvoid, (Ptr{Cvoid},), f.handle)
    f.handle = C_NULL
    nothing
end

function bind!(f::FIRFilter, ::Type{Tx}, nch::Integer) where {Tx}
    if f.handle != C_NULL
        (f.Tx === Tx && f.nchannels == nch) || error("filter is bound to $(f.nchannels) channel(s) of $(f.Tx)")
        return f
    end
    out = Ref{Ptr{Cvoid}}(C_NULL)
    Th = eltype(f.h)
    if f.ratio === nothing && f.polyorder >= 0
        check(ccall((:mrhip_create_farrow, libmr), Cint,
                    (Ptr{Cvoid}, Int64, Cint, Cdouble, Int64, Int64, Cint, Int64, Cint, Ptr{Ptr{Cvoid}}),
                    f.h, length(f.h), dtypecode(Th), f.rate, f.Nphi, f.polyorder, dtypecode(Tx), nch, f.device, out))
    elseif f.ratio === nothing
        check(ccall((:mrhip_create_arbitrary, libmr), Cint,
                    (Ptr{Cvoid}, Int64, Cint, Cdouble, Int64, Cint, Int64, Cint, Ptr{Ptr{Cvoid}}),
                    f.h, length(f.h), dtypecode(Th), f.rate, f.Nphi, dtypecode(Tx), nch, f.device, out))
    else
        check(ccall((:mrhip_create_rational, libmr), Cint,
                    (Ptr{Cvoid}, Int64, Cint, Int64, Int64, Cint, Int64, Cint, Ptr{Ptr{Cvoid}}),
                    f.h, length(f.h), dtypecode(Th), numerator(f.ratio), denominator(f.ratio), dtypecode(Tx), nch,
                    f.device, out))
    end
    f.handle, f.Tx, f.nchannels = out[], Tx, nch
    f
end

function state(f::FIRFilter)
    st = Ref{MRHIPState}()
    check(ccall((:mrhip_get_state, libmr), Cint, (Ptr{Cvoid}, Ptr{MRHIPState}), f.handle, st))
    st[]
end

# `self.kernel` -- the reference's filter object exposes its kernel struct, and its own example reads and writes the
# streaming fields through it (examples/FIRFarrow.jl:23,29: `kernel = myfilter.kernel; kernel.inputDeficit += throwaway`).
# Here the state lives behind the C ABI, so `f.kernel` is a proxy whose fields -- the reference's names, src/Filters.jl:15-147 --
# read mrhip_get_state and write mrhip_set_state.  (Type the Unicode names as in the reference: \itphi<tab>Idx, \alpha<tab>.)
struct KernelProxy{Tk<:FIRKernel}
    filter::FIRFilter{Tk}
end
function Base.getproperty(f::FIRFilter, name::Symbol)
    name === :kernel ? KernelProxy(f) : getfield(f, name)
end
Base.propertynames(::FIRFilter) = (fieldnames(FIRFilter)..., :kernel)
function Base.getproperty(k::KernelProxy, name::Symbol)
    f = getfield(k, :filter)
    name === :filter && return f
    getfield(f, :handle) == C_NULL && error("the kernel fields need a bound filter (call filt once)")
    st = state(f)
    name === :inputDeficit ? Int(st.inputDeficit) :
    name === Symbol("𝜙Idx") ? (f isa FIRFilter{<:FIRFarrow} ? st.phiAccumulator : Int(st.phiIdx)) :     # FIRFarrow.𝜙Idx is the Float64 phase (:131)
    name === Symbol("𝜙Accumulator") ? st.phiAccumulator :
    name === Symbol("α") ? st.alpha :
    name === Symbol("Δ") ? st.delta :
    name === :xIdx ? Int(st.xIdx) :
    name === :rate ? st.rate :
    name === Symbol("N𝜙") ? Int(st.Nphi) :
    name === Symbol("tapsPer𝜙") ? Int(st.tapsPerPhi) :
    name === :hLen ? Int(st.hLen) :
    name === :interpolation ? Int(st.interpolation) :
    name === :decimation ? Int(st.decimation) :
    name === :ratio ? getfield(f, :ratio) :
    error("the kernel has no field $(name)")
end
function Base.setproperty!(k::KernelProxy, name::Symbol, v)
    f = getfield(k, :filter)
    getfield(f, :handle) == C_NULL && error("the kernel fields need a bound filter (call filt once)")
    st = state(f)
    if name === :inputDeficit
        setstate!(f, st.phiIdx, Int(v), st.phiAccumulator)
    elseif name === Symbol("𝜙Idx")
        f isa FIRFilter{<:FIRFarrow} ? setstate!(f, 1, st.inputDeficit, Float64(v)) : setstate!(f, Int(v), st.inputDeficit, st.phiAccumulator)
    elseif name === Symbol("𝜙Accumulator")
        setstate!(f, st.phiIdx, st.inputDeficit, Float64(v))
    elseif name === Symbol("α")                        # 𝜙Accumulator = 𝜙Idx + α (Filters.jl:671-672)
        setstate!(f, st.phiIdx, st.inputDeficit, floor(st.phiAccumulator) + Float64(v))
    else
        error("the kernel field $(name) cannot be set")
    end
    v
end

# ---- bookkeeping ---------------------------------------------------------------------------------
# taps2pfb(h, Nphi)                                        src/Filters.jl:284-298
function taps2pfb(h::Vector{T}, Nphi::Integer) where {T<:Union{Float32,Float64}}
    t = ccall((:mrhip_taps2pfb, libmr), Int64, (Ptr{Cvoid}, Int64, Cint, Int64, Ptr{Cvoid}), h, length(h), dtypecode(T), Nphi, C_NULL)
    pfb = Matrix{T}(undef, t, Nphi)                       # column-major tapsPerPhi x Nphi, as the reference's
    ccall((:mrhip_taps2pfb, libmr), Int64, (Ptr{Cvoid}, Int64, Cint, Int64, Ptr{Cvoid}), h, length(h), dtypecode(T), Nphi, pfb)
    pfb
end
# nextphase(currentphase, ratio)                           src/Filters.jl:433-439
nextphase(p::Integer, ratio::Rational) =
    Int(ccall((:mrhip_nextphase, libmr), Int64, (Int64, Int64, Int64), p, numerator(ratio), denominator(ratio)))
# outputlength / inputlength                               src/Filters.jl:352-422
outputlength(f::FIRFilter, n::Integer) = Int(ccall((:mrhip_outputlength, libmr), Int64, (Ptr{Cvoid}, Int64), f.handle, n))
inputlength(f::FIRFilter, n::Integer) = Int(ccall((:mrhip_inputlength, libmr), Int64, (Ptr{Cvoid}, Int64), f.handle, n))
nextoutputcount(f::FIRFilter, n::Integer) = Int(ccall((:mrhip_next_output_count, libmr), Int64, (Ptr{Cvoid}, Int64), f.handle, n))
# advance the stream state as filt! over n samples would, without data (enter a stream at any sample: time-axis sharding)
advancestate!(f::FIRFilter, n::Integer) = Int(ccall((:mrhip_advance_state, libmr), Int64, (Ptr{Cvoid}, Int64), f.handle, n))
# reset(self::FIRFilter)                                   src/Filters.jl:256-260
function reset(f::FIRFilter)
    f.handle == C_NULL || check(ccall((:mrhip_reset, libmr), Cint, (Ptr{Cvoid},), f.handle))
    f
end

# update()'s mod() (src/Filters.jl:668) as Julia Base before 0.4 computed it for floats (rem(y + rem(x, y), y)) instead of the exact
# remainder: the reference is Julia-0.3 code; identical for a power-of-two N𝜙 (mrhip_set_mod_form)
setmodform!(f::FIRFilter{Tk}, julia03::Bool) where {Tk<:Union{FIRArbitrary,FIRFarrow}} =
    check(ccall((:mrhip_set_mod_form, libmr), Cint, (Ptr{Cvoid}, Cint), f.handle, julia03 ? 1 : 0))

# polyfit(y, polyorder)                                    src/support.jl:85-88 (coefficients, ascending powers)
function polyfit(y::AbstractVector, polyorder::Integer)
    yd = Vector{Float64}(y)
    coef = Vector{Float64}(undef, polyorder + 1)
    check(ccall((:mrhip_polyfit, libmr), Cint, (Ptr{Cdouble}, Int64, Int64, Ptr{Cdouble}), yd, length(yd), polyorder, coef))
    coef
end
# tapsforphase(kernel::FIRFarrow, phase)                   src/Filters.jl:764-775
function tapsforphase(f::FIRFilter{<:FIRFarrow}, phase::Real)
    f.handle == C_NULL && error("tapsforphase needs a bound filter (call filt once)")
    taps = Vector{eltype(f.h)}(undef, state(f).tapsPerPhi)
    check(ccall((:mrhip_farrow_tapsforphase, libmr), Cint, (Ptr{Cvoid}, Cdouble, Ptr{Cvoid}), f.handle, Float64(phase), taps))
    taps
end
# tapsforphase(kernel::FIRArbitrary, phase)                src/Filters.jl:677-690
function tapsforphase(f::FIRFilter{<:FIRArbitrary}, phase::Real)
    f.handle == C_NULL && error("tapsforphase needs a bound filter (call filt once)")
    taps = Vector{eltype(f.h)}(undef, state(f).tapsPerPhi)
    check(ccall((:mrhip_arbitrary_tapsforphase, libmr), Cint, (Ptr{Cvoid}, Cdouble, Ptr{Cvoid}), f.handle, Float64(phase), taps))
    taps
end
# tapsforphase!(buffer, kernel, phase)                     src/Filters.jl:677-688, :764-773 (exported by the reference, Multirate.jl:36)
function tapsforphase!(buffer::Vector, f::FIRFilter{Tk}, phase::Real) where {Tk<:Union{FIRArbitrary,FIRFarrow}}
    taps = tapsforphase(f, phase)
    length(buffer) >= length(taps) || error("buffer is too small")
    copyto!(buffer, 1, taps, 1, length(taps))
    buffer
end
tapsforphase!(buffer::Vector, k::KernelProxy, phase::Real) = tapsforphase!(buffer, getfield(k, :filter), phase)
tapsforphase(k::KernelProxy, phase::Real) = tapsforphase(getfield(k, :filter), phase)
setphase(k::KernelProxy, phi::Real) = setphase(getfield(k, :filter), phi)          # setphase(kernel, 𝜙), Filters.jl:210-232

# How the phase schedule of a FIRArbitrary / FIRFarrow filter has been evaluated so far (mrhip_schedule_info)
function scheduleinfo(f::FIRFilter{Tk}) where {Tk<:Union{FIRArbitrary,FIRFarrow}}
    v = zeros(Int64, 8)
    check(ccall((:mrhip_schedule_info, libmr), Cint, (Ptr{Cvoid}, Ptr{Int64}, Cint), f.handle, v, 8))
    (device_ok = v[1] != 0, ncand = v[2], nwin = v[3], period = v[4], host_steps = v[5], periodic_steps = v[6], device_pieces = v[7], fallback_pieces = v[8])
end

# setphase(self::FIRFilter, 𝜙), 𝜙 in [0, 1]                src/Filters.jl:210-235.  The reference's methods for
# FIRInterpolator/FIRRational read an undefined variable (:212); implemented with the evident intent
# (𝜙Idx = floor(𝜙*N𝜙)+1 clipped to N𝜙).  FIRFarrow as written (:225-229).  FIRArbitrary: the reference sets 𝜙Idx and α and leaves
# 𝜙Accumulator alone (:217-222) -- its next update() then overwrites both from the untouched accumulator; here the ACCUMULATOR is set
# to 𝜙Idx + α (clamped into [1, N𝜙 + 1)), from which 𝜙Idx and α follow as update() derives them: the phase the call asked for survives.
setstate!(f::FIRFilter, phiIdx::Integer, deficit::Integer, acc::Real) =
    check(ccall((:mrhip_set_state, libmr), Cint, (Ptr{Cvoid}, Int64, Int64, Cdouble), f.handle, phiIdx, deficit, Float64(acc)))
function setphase(f::FIRFilter{Tk}, phi::Real) where {Tk<:Union{FIRInterpolator,FIRRational}}
    @assert 0 <= phi <= 1
    st = state(f); idx = min(floor(Int, phi * st.Nphi) + 1, st.Nphi)
    setstate!(f, idx, st.inputDeficit, 1.0); idx
end
function setphase(f::FIRFilter{<:FIRArbitrary}, phi::Real)
    @assert 0 <= phi <= 1
    st = state(f); (alpha, idx) = modf(phi * st.Nphi)
    setstate!(f, 1, st.inputDeficit, clamp(idx + alpha, 1.0, prevfloat(st.Nphi + 1.0))); (idx, alpha)
end
function setphase(f::FIRFilter{<:FIRFarrow}, phi::Real)
    @assert 0 <= phi <= 1
    st = state(f); acc = phi * (st.Nphi - 1) + 1
    setstate!(f, 1, st.inputDeficit, acc); acc
end

# ---- FIR design (src/FIRDesign.jl), host only --------------------------------------------------------
@enum FIRResponse LOWPASS = 0 BANDPASS = 1 HIGHPASS = 2 BANDSTOP = 3        # src/FIRDesign.jl:7
# kaiserlength(transition, attenuation = 60; samplerate = 1.0) -> (numtaps, β)   src/FIRDesign.jl:18-33
function kaiserlength(transition::Real, attenuation::Real = 60; samplerate = 1.0)
    n = Ref{Int64}(0); b = Ref{Cdouble}(0.0)
    check(ccall((:mrhip_kaiserlength, libmr), Cint, (Cdouble, Cdouble, Cdouble, Ptr{Int64}, Ptr{Cdouble}),
                Float64(transition), Float64(attenuation), Float64(samplerate), n, b))
    (Int(n[]), b[])
end
# the window firdes uses when windowfunction == kaiser (beta taken as is, src/Window.jl:53-58)
function kaiser(n::Integer, beta::Real)
    w = Vector{Float64}(undef, n)
    check(ccall((:mrhip_kaiser, libmr), Cint, (Int64, Cdouble, Ptr{Cdouble}), n, Float64(beta), w))
    w
end
cutoffs(F::Real) = Float64[F]
cutoffs(F::AbstractVector) = Vector{Float64}(F)
# firprototype(numtaps, F; response = LOWPASS)                                   src/FIRDesign.jl:47-66
function firprototype(numtaps::Integer, F::Union{Real,AbstractVector}; response::FIRResponse = LOWPASS)
    f = cutoffs(F)
    n = ccall((:mrhip_firprototype, libmr), Int64, (Int64, Ptr{Cdouble}, Cint, Cint, Ptr{Cdouble}), numtaps, f, length(f), Cint(response), C_NULL)
    n < 0 && error(lasterror())
    out = Vector{Float64}(undef, n)
    ccall((:mrhip_firprototype, libmr), Int64, (Int64, Ptr{Cdouble}, Cint, Cint, Ptr{Cdouble}), numtaps, f, length(f), Cint(response), out)
    out
end
# firdes(numtaps, cutoff, windowfunction; response, samplerate, beta)            src/FIRDesign.jl:76-88
function firdes(numtaps::Integer, cutoff::Union{Real,AbstractVector}, windowfunction::Function = kaiser;
                response::FIRResponse = LOWPASS, samplerate = 1.0, beta = 6.75)
    c = cutoffs(cutoff)
    args = (Int64, Ptr{Cdouble}, Cint, Cint, Cdouble, Cdouble, Ptr{Cdouble}, Ptr{Cdouble})
    n = ccall((:mrhip_firdes, libmr), Int64, args, numtaps, c, length(c), Cint(response), Float64(samplerate), Float64(beta), C_NULL, C_NULL)
    n < 0 && error(lasterror())
    out = Vector{Float64}(undef, n)
    if windowfunction == kaiser
        ccall((:mrhip_firdes, libmr), Int64, args, numtaps, c, length(c), Cint(response), Float64(samplerate), Float64(beta), C_NULL, out)
    else
        w = Vector{Float64}(windowfunction(n))
        ccall((:mrhip_firdes, libmr), Int64, args, numtaps, c, length(c), Cint(response), Float64(samplerate), Float64(beta), w, out)
    end
    out
end
# firdes(cutoff, transitionwidth, stopbandAttenuation = 60; response, samplerate)  src/FIRDesign.jl:90-95
function firdes(cutoff::Union{AbstractFloat,AbstractVector}, transitionwidth::Real, stopbandAttenuation::Real = 60;
                response::FIRResponse = LOWPASS, samplerate = 1.0)
    c = cutoffs(cutoff)
    args = (Ptr{Cdouble}, Cint, Cdouble, Cdouble, Cint, Cdouble, Ptr{Cdouble})
    n = ccall((:mrhip_firdes_kaiser, libmr), Int64, args, c, length(c), Float64(transitionwidth), Float64(stopbandAttenuation), Cint(response), Float64(samplerate), C_NULL)
    n < 0 && error(lasterror())
    out = Vector{Float64}(undef, n)
    ccall((:mrhip_firdes_kaiser, libmr), Int64, args, c, length(c), Float64(transitionwidth), Float64(stopbandAttenuation), Cint(response), Float64(samplerate), out)
    out
end

# ---- the hot path --------------------------------------------------------------------------------
promote_out(::Type{Th}, ::Type{Tx}) where {Th,Tx} = promote_type(Th, Tx)   # Filters.jl:476,522,581,636,746

# filt!(buffer, self, x): one channel (Vector) or one channel per column (Matrix), host memory.
# Returns `buffer` for FIRStandard / FIRInterpolator (:472,:516) and the number of samples written for
# FIRRational / FIRDecimator / FIRArbitrary / FIRFarrow (:574,:630,:741,:838), exactly like the reference.
function filt!(buffer::VecOrMat{Tb}, f::FIRFilter{Tk}, x::VecOrMat{Tx}) where {Tb,Tk,Tx}
    nch = size(x, 2)
    bind!(f, Tx, nch)
    Tb === promote_out(eltype(f.h), Tx) || error("buffer eltype must be $(promote_out(eltype(f.h), Tx))")
    nw = Ref{Int64}(0)
    check(ccall((:mrhip_filt_host, libmr), Cint,
                (Ptr{Cvoid}, Ptr{Cvoid}, Int64, Int64, Ptr{Cvoid}, Int64, Int64, Ptr{Int64}),
                f.handle, x, size(x, 1), size(x, 1), buffer, size(buffer, 1), size(buffer, 1), nw))
    (Tk <: FIRStandard || Tk <: FIRInterpolator) ? buffer : Int(nw[])
end

# filt(self, x): allocate, filt!, trim to the samples written (src/Filters.jl:475,519,577,633,744,841)
function filt(f::FIRFilter{Tk}, x::VecOrMat{Tx}) where {Tk,Tx}
    bind!(f, Tx, size(x, 2))
    Tb = promote_out(eltype(f.h), Tx)
    if Tk <: FIRArbitrary || Tk <: FIRFarrow
        # like the reference (:744-752, :841-849): allocate the outputlength estimate (+2, it is only a guess there) and
        # trim to the count filt! returns; the library pipelines the serial phase recurrence with the kernels
        cap = max(outputlength(f, size(x, 1)), 0) + 2
        buffer = x isa Vector ? Vector{Tb}(undef, cap) : Matrix{Tb}(undef, cap, size(x, 2))
        n = size(x, 1) == 0 ? 0 : filt!(buffer, f, x)
        return x isa Vector ? resize!(buffer, n) : buffer[1:n, :]
    end
    n = max(nextoutputcount(f, size(x, 1)), 0)
    buffer = x isa Vector ? Vector{Tb}(undef, n) : Matrix{Tb}(undef, n, size(x, 2))
    size(x, 1) == 0 || filt!(buffer, f, x)
    buffer
end

# stateless forms, src/Filters.jl:858-873
filt(h::Vector, x::VecOrMat, ratio::Rational = 1//1) = filt(FIRFilter(h, ratio), x)
filt(h::Vector, x::VecOrMat, rate::AbstractFloat, Nphi::Integer = 32) = filt(FIRFilter(h, rate, Nphi), x)
filt(h::Vector, x::VecOrMat, rate::AbstractFloat, Nphi::Integer, polyorder::Integer) = filt(FIRFilter(h, rate, Nphi, polyorder), x)

# Device-resident data (e.g. AMDGPU.jl ROCArray): pass raw device pointers and a HIP stream.
# x and y are (n x nchannels) column-major on the filter's device; returns the per-channel output count.
function filt_device!(f::FIRFilter, yptr::Ptr{Cvoid}, ycap::Integer, ystride::Integer, xptr::Ptr{Cvoid}, xlen::Integer,
                      xstride::Integer, ::Type{Tx}, nch::Integer; stream::Ptr{Cvoid} = C_NULL) where {Tx}
    bind!(f, Tx, nch)
    nw = Ref{Int64}(0)
    check(ccall((:mrhip_filt_device, libmr), Cint,
                (Ptr{Cvoid}, Ptr{Cvoid}, Int64, Int64, Ptr{Cvoid}, Int64, Int64, Ptr{Int64}, Ptr{Cvoid}),
                f.handle, xptr, xlen, xstride, yptr, ycap, ystride, nw, stream))
    Int(nw[])
end

# filt! planned ON THE DEVICE from the device-resident stream state (mrhip_filt_device_async): nothing comes back, the host
# never waits; the same call captured into a HIP graph replays at any fixed chunk size, for every kind.  ycap must be at
# least outputlengthbound(f, xlen); countptr (optional) is device-accessible memory for the per-channel output count.
function filt_device_async!(f::FIRFilter, yptr::Ptr{Cvoid}, ycap::Integer, ystride::Integer, xptr::Ptr{Cvoid}, xlen::Integer,
                            xstride::Integer, ::Type{Tx}, nch::Integer; countptr::Ptr{Int64} = Ptr{Int64}(C_NULL), stream::Ptr{Cvoid} = C_NULL) where {Tx}
    bind!(f, Tx, nch)
    check(ccall((:mrhip_filt_device_async, libmr), Cint,
                (Ptr{Cvoid}, Ptr{Cvoid}, Int64, Int64, Ptr{Cvoid}, Int64, Int64, Ptr{Int64}, Ptr{Cvoid}),
                f.handle, xptr, xlen, xstride, yptr, ycap, ystride, countptr, stream))
    nothing
end
# the second stage of a device-resident chain: x is what `prev`'s latest asynchronous / captured call wrote, its length that call's
# count (on the device only); xlenbound = outputlengthbound(prev, ...) sizes the launch (mrhip_filt_device_chained)
function filt_device_chained!(f::FIRFilter, prev::FIRFilter, yptr::Ptr{Cvoid}, ycap::Integer, ystride::Integer, xptr::Ptr{Cvoid}, xlenbound::Integer,
                              xstride::Integer, ::Type{Tx}, nch::Integer; countptr::Ptr{Int64} = Ptr{Int64}(C_NULL), stream::Ptr{Cvoid} = C_NULL) where {Tx}
    bind!(f, Tx, nch)
    check(ccall((:mrhip_filt_device_chained, libmr), Cint,
                (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Int64, Int64, Ptr{Cvoid}, Int64, Int64, Ptr{Int64}, Ptr{Cvoid}),
                f.handle, prev.handle, xptr, xlenbound, xstride, yptr, ycap, ystride, countptr, stream))
    nothing
end
outputlengthbound(f::FIRFilter, n::Integer) = Int(ccall((:mrhip_outputlength_bound, libmr), Int64, (Ptr{Cvoid}, Int64), f.handle, n))
# wait for the filter's enqueued calls, bring the host-side view of the state up to date; returns the last call's count
function syncstate!(f::FIRFilter)
    nw = Ref{Int64}(0)
    check(ccall((:mrhip_sync_state, libmr), Cint, (Ptr{Cvoid}, Ptr{Int64}), f.handle, nw))
    Int(nw[])
end
# FIRFilter.history from device memory (a halo received over RCCL), asynchronous on `stream`
sethistorydevice!(f::FIRFilter, histptr::Ptr{Cvoid}; stream::Ptr{Cvoid} = C_NULL) =
    check(ccall((:mrhip_set_history_device, libmr), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}), f.handle, histptr, stream))

# n INDEPENDENT FIRFilter objects (each its own phase, deficit, history and call length: README.md:87-141) in ONE launch
# (mrhip_filt_device_multi).  xptrs[i] / yptrs[i]: device pointers, one channel per column, column stride = xlens[i] / ycaps[i].
function filt_device_multi!(fs::Vector{<:FIRFilter}, yptrs::Vector{Ptr{Cvoid}}, ycaps::Vector{Int64}, xptrs::Vector{Ptr{Cvoid}},
                            xlens::Vector{Int64}; stream::Ptr{Cvoid} = C_NULL)
    n = length(fs)
    hs = Ptr{Cvoid}[f.handle for f in fs]
    nw = zeros(Int64, n)
    check(ccall((:mrhip_filt_device_multi, libmr), Cint,
                (Ptr{Ptr{Cvoid}}, Cint, Ptr{Ptr{Cvoid}}, Ptr{Int64}, Ptr{Ptr{Cvoid}}, Ptr{Int64}, Ptr{Int64}, Ptr{Cvoid}),
                hs, n, xptrs, xlens, yptrs, ycaps, nw, stream))
    nw
end

# The streaming loop `for each chunk: filt!(view(y, k+1:...), f, view(x, a+1:a+chunk))` over a device-resident signal,
# issued by the library in one call (mrhip_filt_device_chunked): same outputs, same end state.
function filt_device_chunked!(f::FIRFilter, yptr::Ptr{Cvoid}, ycap::Integer, ystride::Integer, xptr::Ptr{Cvoid}, xlen::Integer,
                              xstride::Integer, chunk::Integer, ::Type{Tx}, nch::Integer; stream::Ptr{Cvoid} = C_NULL) where {Tx}
    bind!(f, Tx, nch)
    nw = Ref{Int64}(0)
    check(ccall((:mrhip_filt_device_chunked, libmr), Cint,
                (Ptr{Cvoid}, Ptr{Cvoid}, Int64, Int64, Int64, Ptr{Cvoid}, Int64, Int64, Ptr{Int64}, Ptr{Cvoid}),
                f.handle, xptr, xlen, xstride, chunk, yptr, ycap, ystride, nw, stream))
    Int(nw[])
end

# Device arrays with the strided-array interface (AMDGPU.jl's ROCArray: `pointer`, `size`, `stride`, `eltype`), one
# channel per column: filt!(buffer, f, x) without copies, asynchronous on `stream`.  Kept generic so that this file
# does not depend on AMDGPU.jl; host Arrays keep the methods above.
devptr(A) = Ptr{Cvoid}(UInt(pointer(A)))
colstride(A) = ndims(A) > 1 ? stride(A, 2) : size(A, 1)
function filt_device!(buffer, f::FIRFilter, x; stream::Ptr{Cvoid} = C_NULL)
    stride(x, 1) == 1 && stride(buffer, 1) == 1 || error("x and buffer must be contiguous along time (one channel per column)")
    filt_device!(f, devptr(buffer), size(buffer, 1), colstride(buffer), devptr(x), size(x, 1), colstride(x), eltype(x), size(x, 2); stream = stream)
end
function filt_device_chunked!(buffer, f::FIRFilter, x, chunk::Integer; stream::Ptr{Cvoid} = C_NULL)
    stride(x, 1) == 1 && stride(buffer, 1) == 1 || error("x and buffer must be contiguous along time (one channel per column)")
    filt_device_chunked!(f, devptr(buffer), size(buffer, 1), colstride(buffer), devptr(x), size(x, 1), colstride(x), chunk, eltype(x), size(x, 2); stream = stream)
end

# ---- a ring of arriving chunks (mrhip_ring_*): the streaming loop `for x_i in chunks; y_i = filt(f, x_i); end` (README.md:87-141) fed to ONE
# resident kernel instead of one launch per chunk.  push! is filt!(y, f, x) for the next chunk on device arrays: it returns the
# per-channel output count (the Int the reference's filt! returns) and the chunk's number at once; wait / drain block until outputs are
# complete; close hands the stream (state, history) back to `f`.  x must be complete on the device when pushed.
mutable struct ChunkRing
    filter::FIRFilter
    handle::Ptr{Cvoid}
end
function ChunkRing(f::FIRFilter, ::Type{Tx}, nch::Integer = 1) where {Tx}
    bind!(f, Tx, nch)
    out = Ref{Ptr{Cvoid}}(C_NULL)
    check(ccall((:mrhip_ring_open, libmr), Cint, (Ptr{Cvoid}, Ptr{Ptr{Cvoid}}), f.handle, out))
    r = ChunkRing(f, out[])
    finalizer(close, r)
end
function Base.push!(r::ChunkRing, buffer, x)
    stride(x, 1) == 1 && stride(buffer, 1) == 1 || error("x and buffer must be contiguous along time (one channel per column)")
    nw = Ref{Int64}(0); seq = Ref{UInt64}(0)
    check(ccall((:mrhip_ring_push, libmr), Cint,
                (Ptr{Cvoid}, Ptr{Cvoid}, Int64, Int64, Ptr{Cvoid}, Int64, Int64, Ptr{Int64}, Ptr{UInt64}),
                r.handle, devptr(x), size(x, 1), colstride(x), devptr(buffer), size(buffer, 1), colstride(buffer), nw, seq))
    (Int(nw[]), seq[])
end
# the library's loop of push! over consecutive `chunk`-sample pieces of a device-resident signal, outputs back to back in `buffer`
function pushchunks!(r::ChunkRing, buffer, x, chunk::Integer)
    nw = Ref{Int64}(0); seq = Ref{UInt64}(0)
    check(ccall((:mrhip_ring_push_chunks, libmr), Cint,
                (Ptr{Cvoid}, Ptr{Cvoid}, Int64, Int64, Int64, Ptr{Cvoid}, Int64, Int64, Ptr{Int64}, Ptr{UInt64}),
                r.handle, devptr(x), size(x, 1), colstride(x), chunk, devptr(buffer), size(buffer, 1), colstride(buffer), nw, seq))
    (Int(nw[]), seq[])
end
Base.wait(r::ChunkRing, seq::Integer) = check(ccall((:mrhip_ring_wait, libmr), Cint, (Ptr{Cvoid}, UInt64), r.handle, UInt64(seq)))
drain(r::ChunkRing) = check(ccall((:mrhip_ring_drain, libmr), Cint, (Ptr{Cvoid},), r.handle))
function Base.close(r::ChunkRing)
    r.handle == C_NULL || check(ccall((:mrhip_ring_close, libmr), Cint, (Ptr{Cvoid},), r.handle))
    r.handle = C_NULL
    nothing
end
function ringinfo(r::ChunkRing)
    v = zeros(Int64, 9)
    check(ccall((:mrhip_ring_info, libmr), Cint, (Ptr{Cvoid}, Ptr{Int64}, Cint), r.handle, v, 9))
    (resident = v[1] != 0, depth = v[2], pushed = v[3], restarts = v[4], steps_per_grab = v[5], outputs_per_step = v[6],
     shrunk = v[7], workgroups = v[8], xcds = v[9])
end

# ---- one FIRFilter whose channels are split over several GPUs (mrhip_sharded_*; BASELINE config 5 from Julia) ----------------------
# filt(f, X::Matrix) with one channel per column: the columns are split contiguously over `devices`, every device filters its
# share at the same time (no exchange: channels are independent), the result comes back as one Matrix.  The seam it replaces is
# filt(self, x), src/Filters.jl:577-587, applied column by column with one FIRFilter per column.
mutable struct ShardedFIRFilter
    h::Vector
    nchannels::Int
    devices::Vector{Cint}
    Tx::DataType
    handle::Ptr{Cvoid}
end
function sharded(ctor::Integer, h::Vector{Th}, num, den, rate, Nphi, polyorder, ::Type{Tx}, nch::Integer, devices) where {Th<:Union{Float32,Float64},Tx}
    devs = Cint[d for d in devices]
    out = Ref{Ptr{Cvoid}}(C_NULL)
    check(ccall((:mrhip_sharded_create, libmr), Cint,
                (Cint, Ptr{Cvoid}, Int64, Cint, Int64, Int64, Cdouble, Int64, Int64, Cint, Int64, Ptr{Cint}, Cint, Ptr{Ptr{Cvoid}}),
                ctor, h, length(h), dtypecode(Th), num, den, Float64(rate), Nphi, polyorder, dtypecode(Tx), nch, devs, length(devs), out))
    s = ShardedFIRFilter(copy(h), nch, devs, Tx, out[])
    finalizer(s -> (s.handle == C_NULL || ccall((:mrhip_sharded_destroy, libmr), Cvoid, (Ptr{Cvoid},), s.handle); s.handle = C_NULL), s)
end
# the reference's three constructors (src/Filters.jl:158,183,192) + sample type, channel count and the devices
ShardedFIRFilter(h::Vector, ratio::Rational, ::Type{Tx}, nch::Integer, devices) where {Tx} =
    sharded(0, h, numerator(ratio), denominator(ratio), 0.0, 0, 0, Tx, nch, devices)
ShardedFIRFilter(h::Vector, rate::AbstractFloat, Nphi::Integer, ::Type{Tx}, nch::Integer, devices) where {Tx} =
    (rate > 0.0 || error("rate must be greater than 0"); sharded(1, h, 1, 1, rate, Nphi, 0, Tx, nch, devices))
ShardedFIRFilter(h::Vector, rate::AbstractFloat, Nphi::Integer, polyorder::Integer, ::Type{Tx}, nch::Integer, devices) where {Tx} =
    (rate > 0.0 || error("rate must be greater than 0"); sharded(2, h, 1, 1, rate, Nphi, polyorder, Tx, nch, devices))
outputlength(s::ShardedFIRFilter, n::Integer) = Int(ccall((:mrhip_sharded_outputlength, libmr), Int64, (Ptr{Cvoid}, Int64), s.handle, n))
reset(s::ShardedFIRFilter) = (check(ccall((:mrhip_sharded_reset, libmr), Cint, (Ptr{Cvoid},), s.handle)); s)
function filt!(buffer::Matrix{Tb}, s::ShardedFIRFilter, X::Matrix{Tx}) where {Tb,Tx}
    Tx === s.Tx && size(X, 2) == s.nchannels || error("X must be a Matrix{$(s.Tx)} with $(s.nchannels) columns")
    Tb === promote_out(eltype(s.h), Tx) && size(buffer, 2) == s.nchannels || error("buffer must be a Matrix{$(promote_out(eltype(s.h), Tx))} with $(s.nchannels) columns")
    nw = Ref{Int64}(0)
    check(ccall((:mrhip_sharded_filt_host, libmr), Cint,
                (Ptr{Cvoid}, Ptr{Cvoid}, Int64, Int64, Ptr{Cvoid}, Int64, Int64, Ptr{Int64}),
                s.handle, X, size(X, 1), size(X, 1), buffer, size(buffer, 1), size(buffer, 1), nw))
    Int(nw[])
end
function filt(s::ShardedFIRFilter, X::Matrix{Tx}) where {Tx}
    buffer = Matrix{promote_out(eltype(s.h), Tx)}(undef, max(outputlength(s, size(X, 1)), 0) + 2, s.nchannels)
    n = size(X, 1) == 0 ? 0 : filt!(buffer, s, X)
    buffer[1:n, :]
end

# ---- cascades (mrhip_cascade_*): stages chained on the device, intermediates resident in HBM -------------------
mutable struct FilterCascade
    stages::Vector{FIRFilter}
    handle::Ptr{Cvoid}
    function FilterCascade(stages::FIRFilter...)
        isempty(stages) && error("FilterCascade takes one or more FIRFilter stages")
        c = new(collect(FIRFilter, stages), C_NULL)
        finalizer(c -> (c.handle == C_NULL || ccall((:mrhip_cascade_destroy, libmr), Cvoid, (Ptr{Cvoid},), c.handle); c.handle = C_NULL), c)
    end
end
function bind!(c::FilterCascade, ::Type{Tx}, nch::Integer) where {Tx}
    if c.handle != C_NULL                   # bound: a later call must have the sample type / channel count of the first
        bind!(c.stages[1], Tx, nch)         # (errors on a mismatch, like a lone FIRFilter)
        return c
    end
    T = Tx
    for f in c.stages                       # stage i+1's sample type is stage i's output type
        bind!(f, T, nch); T = promote_out(eltype(f.h), T)
    end
    out = Ref{Ptr{Cvoid}}(C_NULL)
    hs = Ptr{Cvoid}[f.handle for f in c.stages]
    check(ccall((:mrhip_cascade_create, libmr), Cint, (Ptr{Ptr{Cvoid}}, Cint, Ptr{Ptr{Cvoid}}), hs, length(hs), out))
    c.handle = out[]
    c
end
outputlength(c::FilterCascade, n::Integer) = Int(ccall((:mrhip_cascade_outputlength, libmr), Int64, (Ptr{Cvoid}, Int64), c.handle, n))
nextoutputcount(c::FilterCascade, n::Integer) = Int(ccall((:mrhip_cascade_next_output_count, libmr), Int64, (Ptr{Cvoid}, Int64), c.handle, n))
reset(c::FilterCascade) = (c.handle == C_NULL || check(ccall((:mrhip_cascade_reset, libmr), Cint, (Ptr{Cvoid},), c.handle)); c)
# the chain with nothing returned to the host (asynchronous / capturable at any chunk size): the first stage is planned on the device,
# every later one takes its input length from the previous stage's count on the device (mrhip_cascade_filt_device_async)
function filt_device_async!(buffer, c::FilterCascade, x; countptr::Ptr{Int64} = Ptr{Int64}(C_NULL), stream::Ptr{Cvoid} = C_NULL)
    bind!(c, eltype(x), size(x, 2))
    check(ccall((:mrhip_cascade_filt_device_async, libmr), Cint,
                (Ptr{Cvoid}, Ptr{Cvoid}, Int64, Int64, Ptr{Cvoid}, Int64, Int64, Ptr{Int64}, Ptr{Cvoid}),
                c.handle, devptr(x), size(x, 1), colstride(x), devptr(buffer), size(buffer, 1), colstride(buffer), countptr, stream))
    nothing
end
# filt!(buffer, cascade, x) on device arrays: returns the per-channel output count
function filt_device!(buffer, c::FilterCascade, x; stream::Ptr{Cvoid} = C_NULL)
    bind!(c, eltype(x), size(x, 2))
    nw = Ref{Int64}(0)
    check(ccall((:mrhip_cascade_filt_device, libmr), Cint,
                (Ptr{Cvoid}, Ptr{Cvoid}, Int64, Int64, Ptr{Cvoid}, Int64, Int64, Ptr{Int64}, Ptr{Cvoid}),
                c.handle, devptr(x), size(x, 1), colstride(x), devptr(buffer), size(buffer, 1), colstride(buffer), nw, stream))
    Int(nw[])
end
# host arrays: the stages' host path one after the other (what a user of the reference writes by hand)
filt(c::FilterCascade, x::VecOrMat) = foldl((sig, f) -> filt(f, sig), c.stages; init = x)

end # module
