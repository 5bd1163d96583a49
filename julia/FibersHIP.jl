#=
  FibersHIP.jl — the binding a Fibers.jl maintainer would add to route the hot path through
  libfibers_hip.so (MI355X / gfx950).  Mechanically derived from include/fibers_hip.h.

  NOT EXECUTED IN THIS REPOSITORY'S CI: the build image has no Julia.  It keeps the reference's
  signatures (dti_fit, adc_fit, gqi_rec, dsi_rec, stream) and result structs; only the bodies change.
  Include after src/Fibers.jl's own definitions of MRI, ODF, DTI, GQI, DSI, Tract, str_add!.
=#

const libfibers = get(ENV, "FIBERS_HIP_LIB", "libfibers_hip.so")

const FIB_DTYPE = Dict(UInt8=>0, Int8=>1, Int16=>2, UInt16=>3, Int32=>4, UInt32=>5,
                       Float32=>6, Float64=>7, Int64=>8, Bool=>9)
# OR-ed into mask_dtype by the fits below: their outputs are MRI(mask, n, Float32) = zeros (mri.jl:251-255), so the library need not
# write the voxels outside the mask (include/fibers_hip.h)
const FIB_MASK_OUTPUTS_ZEROED = Cint(0x100)

# Multi-GPU: `device = FIB_DEVICE_ALL` shards a call over the device set declared here (contiguous voxel slabs for the fits,
# as Threads.@threads shards the z loop in dti.jl:258 / gqi.jl:132 / dsi.jl:197; round-robin seeds for stream).  Without
# fib_init the set is every visible GPU.  Results do not depend on the set.
const FIB_DEVICE_ALL = Cint(-1)
fib_init(devs::Vector{<:Integer}=Int[]) = fib_check(ccall((:fib_init, libfibers), Cint, (Cint, Ptr{Cint}), length(devs), Cint.(devs)))
fib_trim() = ccall((:fib_trim, libfibers), Cint, ())          # buffers kept between calls go back to the driver (plans stay)
fib_shutdown() = ccall((:fib_shutdown, libfibers), Cvoid, ())

function fib_check(rc::Cint)
  rc == 0 && return
  msg = unsafe_string(ccall((:fib_last_error, libfibers), Cstring, ()))
  error(msg)                       # same strings as the reference's error() calls
end

# layout: fib_dti_out sizeof 80: s0@0 eigval1@8 eigval2@16 eigval3@24 eigvec1@32 eigvec2@40 eigvec3@48 rd@56 md@64 fa@72
struct FibDtiOut
  s0::Ptr{Float32}; eigval1::Ptr{Float32}; eigval2::Ptr{Float32}; eigval3::Ptr{Float32}
  eigvec1::Ptr{Float32}; eigvec2::Ptr{Float32}; eigvec3::Ptr{Float32}
  rd::Ptr{Float32}; md::Ptr{Float32}; fa::Ptr{Float32}
end

"dti_fit(dwi::MRI, mask::MRI) — replaces dti.jl:221-316"
function dti_fit(dwi::MRI, mask::MRI; device::Integer=0)
  isempty(dwi.bval) && error("Missing b-value table from input DWI structure")
  isempty(dwi.bvec) && error("Missing gradient table from input DWI structure")
  nx, ny, nz, nvol = size(dwi.vol)
  S0 = MRI(mask, 1, Float32); E1 = MRI(mask, 1, Float32); E2 = MRI(mask, 1, Float32); E3 = MRI(mask, 1, Float32)
  V1 = MRI(mask, 3, Float32); V2 = MRI(mask, 3, Float32); V3 = MRI(mask, 3, Float32)
  RD = MRI(mask, 1, Float32); MD = MRI(mask, 1, Float32); FA = MRI(mask, 1, Float32)
  vol = dwi.vol::Array{Float32,4}; m = mask.vol
  GC.@preserve vol m S0 E1 E2 E3 V1 V2 V3 RD MD FA begin
    out = Ref(FibDtiOut(pointer(S0.vol), pointer(E1.vol), pointer(E2.vol), pointer(E3.vol),
                        pointer(V1.vol), pointer(V2.vol), pointer(V3.vol),
                        pointer(RD.vol), pointer(MD.vol), pointer(FA.vol)))
    fib_check(ccall((:fib_dti_fit, libfibers), Cint,
                    (Cint, Ptr{Float32}, Cint, Cint, Cint, Cint, Ptr{Cvoid}, Cint, Ptr{Float32}, Ptr{Float32}, Ref{FibDtiOut}),
                    device, vol, nx, ny, nz, nvol, m, FIB_DTYPE[eltype(m)] | FIB_MASK_OUTPUTS_ZEROED, dwi.bval, dwi.bvec, out))
  end
  return DTI(S0, E1, E2, E3, V1, V2, V3, RD, MD, FA)
end

"adc_fit(dwi::MRI, mask::MRI) — replaces dti.jl:164-213"
function adc_fit(dwi::MRI, mask::MRI; device::Integer=0)
  isempty(dwi.bval) && error("Missing b-value table from input DWI structure")
  nx, ny, nz, nvol = size(dwi.vol)
  adc = MRI(mask, 1, Float32); s0 = MRI(mask, 1, Float32)
  vol = dwi.vol::Array{Float32,4}; m = mask.vol
  GC.@preserve vol m adc s0 fib_check(ccall((:fib_adc_fit, libfibers), Cint,
      (Cint, Ptr{Float32}, Cint, Cint, Cint, Cint, Ptr{Cvoid}, Cint, Ptr{Float32}, Ptr{Float32}, Ptr{Float32}),
      device, vol, nx, ny, nz, nvol, m, FIB_DTYPE[eltype(m)] | FIB_MASK_OUTPUTS_ZEROED, dwi.bval, adc.vol, s0.vol))
  return adc, s0
end

"find_peaks(odf, odf_dirs) — find_peaks!(W) (gqi.jl:180-201) for a whole ODF volume [nx,ny,nz,nvert]:
 returns (isort_top [nx,ny,nz,3] 1-based first-half vertex rows, 0 where absent; nvalid [nx,ny,nz])"
function find_peaks(odf::MRI, odf_dirs::ODF=sphere_642; device::Integer=0)
  nx, ny, nz, nvert = size(odf.vol)
  nvox = nx * ny * nz
  top = Array{Int32}(undef, nx, ny, nz, 3); nvalid = Array{Int32}(undef, nx, ny, nz)
  faces = Int32.(odf_dirs.faces); verts = odf_dirs.vertices; vol = odf.vol::Array{Float32,4}
  GC.@preserve vol top nvalid faces verts fib_check(ccall((:fib_find_peaks, libfibers), Cint,
      (Cint, Ptr{Float32}, Int64, Ptr{Float32}, Cint, Ptr{Int32}, Cint, Ptr{Int32}, Ptr{Int32}),
      device, vol, nvox, verts, size(verts, 1), faces, size(faces, 1), top, nvalid))
  return top .+ Int32(1), nvalid
end

"""
    find_peaks!(W::Union{GQIwork, DSIwork})

The reference's own surface (gqi.jl:180-201): reads `W.o[tid]` of the calling thread, fills `W.odf_peak[tid]` (amplitudes of the
local peaks, 0 elsewhere) and `W.isort[tid]` (`sortperm(odf_peak, rev=true)`, 1-based), returns `count(odf_peak .> 0)`.
`W.faces` is the FOLDED face table of the work struct (vertex indices 1..nvert); the library folds faces itself, so the
unfolded tessellation is rebuilt by pairing every half-sphere vertex with a placeholder antipode that no face uses.
One voxel per call, as in the reference; `gqi_rec` / `dsi_rec` do not come through here (they find the peaks on the GPU while
the ODF is on chip) — this is for code that calls `find_peaks!` directly.
"""
function find_peaks!(W; device::Integer=0)
  tid = Threads.threadid()
  o = W.o[tid]::Vector{Float32}
  nvert = W.nvert
  faces = Int32.(W.faces)                                 # folded, 1-based, [nf x 3]
  verts = zeros(Float32, 2 * nvert, 3)                    # coordinates are not used by find_peaks!
  pk = Vector{Float32}(undef, nvert); isort = Vector{Int32}(undef, nvert); nvalid = Ref{Int32}(0)
  GC.@preserve o pk isort faces verts fib_check(ccall((:fib_find_peaks_work, libfibers), Cint,
      (Cint, Ptr{Float32}, Int64, Ptr{Float32}, Cint, Ptr{Int32}, Cint, Ptr{Float32}, Ptr{Int32}, Ref{Int32}),
      device, o, 1, verts, 2 * nvert, faces, size(faces, 1), pk, isort, nvalid))
  W.odf_peak[tid] .= pk
  W.isort[tid] .= Int.(isort) .+ 1
  return Int(nvalid[])
end

"gqi_rec(dwi, mask, odf_dirs, σ) — replaces gqi.jl:109-171 (and find_peaks! gqi.jl:180-201)"
function gqi_rec(dwi::MRI, mask::MRI, odf_dirs::ODF=sphere_642, σ::Float32=Float32(1.25); device::Integer=0)
  isempty(dwi.bval) && error("Missing b-value table from input DWI structure")
  isempty(dwi.bvec) && error("Missing gradient table from input DWI structure")
  nx, ny, nz, nvol = size(dwi.vol)
  nvert = div(size(odf_dirs.vertices, 1), 2)
  odf = MRI(mask, nvert, Float32)
  peak = [MRI(mask, 3, Float32) for _ in 1:3]; qa = [MRI(mask, 1, Float32) for _ in 1:3]
  faces = Int32.(odf_dirs.faces); verts = odf_dirs.vertices; vol = dwi.vol::Array{Float32,4}; m = mask.vol
  pk = [pointer(p.vol) for p in peak]; pq = [pointer(q.vol) for q in qa]
  GC.@preserve vol m odf peak qa faces verts fib_check(ccall((:fib_gqi_rec, libfibers), Cint,
      (Cint, Ptr{Float32}, Cint, Cint, Cint, Cint, Ptr{Cvoid}, Cint, Ptr{Float32}, Ptr{Float32},
       Ptr{Float32}, Cint, Ptr{Int32}, Cint, Cfloat, Ptr{Float32}, Ptr{Ptr{Float32}}, Ptr{Ptr{Float32}}),
      device, vol, nx, ny, nz, nvol, m, FIB_DTYPE[eltype(m)] | FIB_MASK_OUTPUTS_ZEROED, dwi.bval, dwi.bvec,
      verts, size(verts, 1), faces, size(faces, 1), σ, odf.vol, pk, pq))
  return GQI(odf, peak, qa)
end

"dsi_rec(dwi, mask, odf_dirs, hann_width) — replaces dsi.jl:171-270"
function dsi_rec(dwi::MRI, mask::MRI, odf_dirs::ODF=sphere_642, hann_width::Int=32; device::Integer=0)
  isempty(dwi.bval) && error("Missing b-value table from input DWI structure")
  isempty(dwi.bvec) && error("Missing gradient table from input DWI structure")
  nx, ny, nz, nvol = size(dwi.vol)
  nvert = div(size(odf_dirs.vertices, 1), 2)
  pdf = MRI(mask, nvol, Float32); odf = MRI(mask, nvert, Float32)
  peak = [MRI(mask, 3, Float32) for _ in 1:3]; qa = [MRI(mask, 1, Float32) for _ in 1:3]
  faces = Int32.(odf_dirs.faces); verts = odf_dirs.vertices; vol = dwi.vol::Array{Float32,4}; m = mask.vol
  pk = [pointer(p.vol) for p in peak]; pq = [pointer(q.vol) for q in qa]
  GC.@preserve vol m pdf odf peak qa faces verts fib_check(ccall((:fib_dsi_rec, libfibers), Cint,
      (Cint, Ptr{Float32}, Cint, Cint, Cint, Cint, Ptr{Cvoid}, Cint, Ptr{Float32}, Ptr{Float32},
       Ptr{Float32}, Cint, Ptr{Int32}, Cint, Cint, Ptr{Float32}, Ptr{Float32}, Ptr{Ptr{Float32}}, Ptr{Ptr{Float32}}),
      device, vol, nx, ny, nz, nvol, m, FIB_DTYPE[eltype(m)] | FIB_MASK_OUTPUTS_ZEROED, dwi.bval, dwi.bvec,
      verts, size(verts, 1), faces, size(faces, 1), hann_width, pdf.vol, odf.vol, pk, pq))
  return DSI(pdf, odf, peak, qa)
end

# layout: fib_rumba_out sizeof 80: fodf@0 fgm@8 fcsf@16 gfa@24 var@32 peak@40
struct FibRumbaOut
  fodf::Ptr{Float32}; fgm::Ptr{Float32}; fcsf::Ptr{Float32}; gfa::Ptr{Float32}; var::Ptr{Float32}
  peak::NTuple{5, Ptr{Float32}}
end

"st_eigen(Sxx, Sxy, Sxz, Syy, Syz, Szz) — replaces structens.jl:13-37"
function st_eigen(Sxx::Array{Float32,3}, Sxy::Array{Float32,3}, Sxz::Array{Float32,3},
                  Syy::Array{Float32,3}, Syz::Array{Float32,3}, Szz::Array{Float32,3}; device::Integer=0)
  eigvec = Array{Float32,5}(undef, size(Sxx)..., 3, 3)
  eigval = Array{Float32,4}(undef, size(Sxx)..., 3)
  S = [pointer(Sxx), pointer(Sxy), pointer(Sxz), pointer(Syy), pointer(Syz), pointer(Szz)]
  GC.@preserve Sxx Sxy Sxz Syy Syz Szz S eigvec eigval fib_check(ccall((:fib_st_eigen, libfibers), Cint,
      (Cint, Ptr{Ptr{Cfloat}}, Int64, Ptr{Cfloat}, Ptr{Cfloat}), device, S, length(Sxx), eigvec, eigval))
  return eigvec, eigval
end

"rumba_rec(dwi, mask, odf_dirs, niter, ...) — replaces rusd.jl:419-636"
function rumba_rec(dwi::MRI, mask::MRI, odf_dirs::ODF=sphere_724, niter::Integer=600, λ_para::Float32=Float32(1.7e-3),
                   λ_perp::Float32=Float32(0.2e-3), λ_csf::Float32=Float32(3.0e-3), λ_gm::Float32=Float32(0.8e-4),
                   ncoils::Integer=1, coil_combine::String="SMF-SENSE", ipat_factor::Integer=1, use_tv::Bool=true;
                   device::Integer=0)
  isempty(dwi.bval) && error("Missing b-value table from input DWI structure")
  isempty(dwi.bvec) && error("Missing gradient table from input DWI structure")
  sos = coil_combine == "SoS-GRAPPA" ? 1 : (coil_combine == "SMF-SENSE" ? 0 : error("Unknown coil combine mode " * coil_combine))
  ipat_factor < 1 && error("iPAT factor must be a positive integer")
  nx, ny, nz, nvol = size(dwi.vol)
  nvert = div(size(odf_dirs.vertices, 1), 2)
  fodf = MRI(mask, nvert, Float32); fgm = MRI(mask, 1, Float32); fcsf = MRI(mask, 1, Float32)
  gfa = MRI(mask, 1, Float32); var = MRI(mask, 1, Float32); peak = [MRI(mask, 3, Float32) for _ in 1:5]
  verts = odf_dirs.vertices; vol = dwi.vol::Array{Float32,4}; m = mask.vol
  snr = Ref{Float32}(0); snrsd = Ref{Float32}(0)
  out = Ref(FibRumbaOut(pointer(fodf.vol), pointer(fgm.vol), pointer(fcsf.vol), pointer(gfa.vol), pointer(var.vol),
                        ntuple(i -> pointer(peak[i].vol), 5)))
  GC.@preserve vol m fodf fgm fcsf gfa var peak verts fib_check(ccall((:fib_rumba_rec, libfibers), Cint,
      (Cint, Ptr{Float32}, Cint, Cint, Cint, Cint, Ptr{Cvoid}, Cint, Ptr{Float32}, Ptr{Float32}, Ptr{Float32}, Cint, Cint,
       Cfloat, Cfloat, Cfloat, Cfloat, Cint, Cint, Cint, Cint, Ref{FibRumbaOut}, Ref{Float32}, Ref{Float32}),
      device, vol, nx, ny, nz, nvol, m, FIB_DTYPE[eltype(m)], dwi.bval, dwi.bvec, verts, size(verts, 1), niter,
      λ_para, λ_perp, λ_csf, λ_gm, ncoils, sos, ipat_factor, use_tv ? 1 : 0, out, snr, snrsd))
  return RUMBASD(fodf, fgm, fcsf, peak, gfa, var, snr[], snrsd[])
end

# layout: fib_stream_params sizeof 64: nx@0 ny@4 nz@8 nvec@12 len_min@16 len_max@20 cosang_thresh@24 step_size@28 smooth_coeff@32 search_dist@36 search_cosang@40 ws@48 interp@56 search_flat_axis@60
struct FibStreamParams
  nx::Int32; ny::Int32; nz::Int32; nvec::Int32; len_min::Int32; len_max::Int32
  cosang_thresh::Float32; step_size::Float32; smooth_coeff::Float32
  search_dist::Int32; search_cosang::Float32          # microscopy regime (stream.jl:83, 547-619) when search_dist > 0
  ws::Ptr{Cvoid}                                      # optional tracer workspace (fibd_stream_ws_create); C_NULL for the host-buffer calls
  interp::Int32                                       # 0: nearest voxel (stream.jl:514); 1: trilinear blend (not in the reference)
  search_flat_axis::Int32                             # microscopy regime + 2-D angle inputs: 1..3 = the through-plane axis (search distance 0, stream.jl:153-155); 0: none
end

# layout: fib_tract_out sizeof 48: nlines@0 npoints@8 npts@16 seed_index@24 xyz@32 flags@40
mutable struct FibTractOut
  nlines::Int64; npoints::Int64
  npts::Ptr{Int32}; seed_index::Ptr{Int64}; xyz::Ptr{Float32}; flags::Ptr{UInt8}
  FibTractOut() = new(0, 0, C_NULL, C_NULL, C_NULL, C_NULL)
end

"stream(ovec; ...) — replaces stream.jl:730-790 for the angle-picking path and the microscopy regime (no lcms)"
function stream(ovec::Union{MRI,Vector{MRI}}; f::Union{MRI,Vector{MRI},Nothing}=nothing, f_thresh::Real=.03,
                fa::Union{MRI,Nothing}=nothing, fa_thresh::Real=.1, mask::Union{MRI,Nothing}=nothing,
                seed::Union{MRI,Nothing}=nothing, nsub::Union{Integer,Nothing}=3, len_min::Integer=3,
                len_max::Integer=(isa(ovec,MRI) ? maximum(ovec.volsize) : maximum(ovec[1].volsize)),
                ang_thresh::Union{Real,Nothing}=45, step_size::Union{Real,Nothing}=.5,
                smooth_coeff::Union{Real,Nothing}=.2, search_dist::Integer=15, search_ang::Real=10,
                lcms::Union{MRI,Nothing}=nothing, lcm_thresh::Real=.099, rng_seed::Integer=rand(UInt64),
                device::Integer=0)
  ovecs = isa(ovec, MRI) ? MRI[ovec] : ovec
  fs    = isa(f, MRI) ? MRI[f] : f
  nx, ny, nz = size(ovecs[1].vol)[1:3]
  # 2-D orientation angles (one frame) become 3-D vectors here, with the reference's own arithmetic (stream.jl:147-172):
  # through-plane = the dimension with the largest voxel size, cos / sin (radians) or cosd / sind (degrees) in the other two
  flat_axis = Int32(0)
  lcm_frames = size(ovecs[1].vol, 4)                                        # what stream.jl:221 looks at (BEFORE the expansion)
  lcm_zero = [all(x -> x == 0, view(ovecs[1].vol, :, :, :, c)) for c in 1:lcm_frames]
  ovecs = map(ovecs) do o
    size(o.vol, 4) == 3 && return o
    size(o.vol, 4) == 1 || error("Input orientations should be 3D vectors or angles ∊ [-90, 90]")
    thrudim = argmax(o.volres); strdims = setdiff(1:3, thrudim)
    flat_axis = Int32(thrudim)
    a = view(o.vol, :, :, :, 1)
    v = zeros(Float32, nx, ny, nz, 3)
    if -π/2-eps(Float32) <= minimum(a) && maximum(a) <= π/2+eps(Float32)
      v[:, :, :, strdims[1]] .= cos.(a);  v[:, :, :, strdims[2]] .= sin.(a)
    elseif -90 <= minimum(a) && maximum(a) <= 90
      v[:, :, :, strdims[1]] .= cosd.(a); v[:, :, :, strdims[2]] .= sind.(a)
    else
      error("Input orientations should be 3D vectors or angles ∊ [-90, 90]")
    end
    e = MRI(o, 3, Float32); e.vol .= v
    e
  end
  if !isnothing(seed) && size(seed.vol) != size(mask.vol)
    error("Dimension mismatch between seed mask " * string(size(seed.vol)) * " and brain mask " * string(size(mask.vol)))
  end
  domicro = minimum(ovecs[1].volres) <= 0.05                                # stream.jl:83
  isnothing(nsub) && (nsub = domicro ? 0 : 3); isnothing(ang_thresh) && (ang_thresh = domicro ? 20 : 45)   # :89-92
  isnothing(step_size) && (step_size = domicro ? 1 : .5); isnothing(smooth_coeff) && (smooth_coeff = domicro ? 0 : .2)
  # sub-voxel offsets from the GLOBAL RNG, exactly as stream.jl:176-181
  sublist = nsub > 0 ? hcat([Float32.(rand(Uniform(-.5+eps(), .5-eps()), 3)) for _ in 1:nsub]...) : zeros(Float32, 3, 1)
  prm = Ref(FibStreamParams(nx, ny, nz, length(ovecs), len_min, len_max,
                            cosd(Float32(ang_thresh)), Float32(step_size), Float32(smooth_coeff),
                            domicro ? Int32(search_dist) : Int32(0), cosd(Float32(search_ang)), C_NULL, Int32(0),
                            domicro ? flat_axis : Int32(0)))                   # micro_search_dist[thrudim] = 0, stream.jl:153-155
  pv = [pointer(o.vol) for o in ovecs]
  pf = isnothing(fs) ? C_NULL : [pointer(x.vol) for x in fs]
  out = FibTractOut()
  if !isnothing(lcms)     # LCM-guided tracking (stream.jl:380-495); the library's uniform stream replaces the global RNG
    lv = lcms.vol::Array{Float32,4}
    GC.@preserve ovecs fs fa mask seed sublist pv pf lv fib_check(ccall((:fib_stream_lcm, libfibers), Cint,
      (Cint, Ref{FibStreamParams}, Ptr{Ptr{Float32}}, Ptr{Ptr{Float32}}, Cfloat, Ptr{Float32}, Cfloat,
       Ptr{Cvoid}, Cint, Ptr{Cvoid}, Cint, Ptr{Float32}, Cint, Ptr{Float32}, Cfloat, UInt64, Ref{FibTractOut}),
      device, prm, pv, pf, Float32(f_thresh), isnothing(fa) ? C_NULL : pointer(fa.vol), Float32(fa_thresh),
      isnothing(mask) ? C_NULL : pointer(mask.vol), isnothing(mask) ? 0 : FIB_DTYPE[eltype(mask.vol)],
      isnothing(seed) ? C_NULL : pointer(seed.vol), isnothing(seed) ? 0 : FIB_DTYPE[eltype(seed.vol)],
      sublist, size(sublist, 2), lv, Float32(lcm_thresh), UInt64(rng_seed), out))
  else
  GC.@preserve ovecs fs fa mask seed sublist pv pf fib_check(ccall((:fib_stream, libfibers), Cint,
      (Cint, Ref{FibStreamParams}, Ptr{Ptr{Float32}}, Ptr{Ptr{Float32}}, Cfloat, Ptr{Float32}, Cfloat,
       Ptr{Cvoid}, Cint, Ptr{Cvoid}, Cint, Ptr{Float32}, Cint, Ref{FibTractOut}),
      device, prm, pv, pf, Float32(f_thresh), isnothing(fa) ? C_NULL : pointer(fa.vol), Float32(fa_thresh),
      isnothing(mask) ? C_NULL : pointer(mask.vol), isnothing(mask) ? 0 : FIB_DTYPE[eltype(mask.vol)],
      isnothing(seed) ? C_NULL : pointer(seed.vol), isnothing(seed) ? 0 : FIB_DTYPE[eltype(seed.vol)],
      sublist, size(sublist, 2), out))
  end
  npts = unsafe_wrap(Array, out.npts, out.nlines)
  xyz  = unsafe_wrap(Array, out.xyz, (3, Int(out.npoints)))
  off  = cumsum(vcat(0, Int.(npts)))
  str  = [xyz[:, off[i]+1:off[i+1]] for i in 1:length(npts)]          # Vector{Matrix{Float32}} [3 x npts]
  flag = isnothing(lcms) ? nothing :
         (fl = unsafe_wrap(Array, out.flags, Int(out.npoints)); [Float32.(fl[off[i]+1:off[i+1]]) for i in 1:length(npts)])
  ccall((:fib_tract_free, libfibers), Cvoid, (Ref{FibTractOut},), out)
  tr = Tract{Float32}(mask)
  str_add!(tr, str, flag)                                             # stream.jl:784-787
  return tr
end
