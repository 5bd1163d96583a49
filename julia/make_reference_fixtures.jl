#=
  make_reference_fixtures.jl — the pinning hook: runs THE REFERENCE (lincbrain/Fibers.jl) on the exact inputs of this
  repository's golden fixtures and writes its outputs next to them, so that tests/test_reference_fixtures.py can compare the
  oracle and the HIP path with the reference itself instead of with a restatement of it.

  NOT EXECUTED IN THIS REPOSITORY'S CI: the build image has no Julia and the reference ships no fixtures of its own
  (test/runtests.jl:4-6 is empty).  Until someone runs this file, parity is "unpinned" (DESIGN.md §5).

  usage (on a machine with Julia >= 1.7 and the reference checked out):
      python tests/golden/export_raw.py                                    # inputs -> tests/golden/raw/<case>/
      julia --threads 1 julia/make_reference_fixtures.jl /path/to/Fibers.jl tests/golden/raw tests/golden/reference
      python -m pytest tests/test_reference_fixtures.py                    # (+ -m gpu on an MI355X box)

  --threads 1: the reference's work structs are indexed by Threads.threadid(); with one thread the streamline order is the
  seed order (stream.jl:757-787) without relying on the static schedule.

  Exchange format (tests/golden/refio.py): per case a directory with `<key>.bin` (raw little-endian, column-major) and
  `meta.txt` (`<key> <dtype> <ndim> <dims...>` per array, `<key> = <value>` per scalar).  No package beyond the reference.

  What is called, by fixture kind:
    dti    dti_fit(dwi, mask) (dti.jl:221), adc_fit(dwi, mask) (dti.jl:164)
    gqi    gqi_rec(dwi, mask, odf_dirs, σ) (gqi.jl:109)
    dsi    dsi_rec(dwi, mask, sphere_642, hann_width) (dsi.jl:171)
    peaks  find_peaks!(W) (gqi.jl:180) on a GQIwork whose amplitudes are set to each fixture column
    stream / micro
           StreamWork(...) (stream.jl:74) with its sub-voxel offsets REPLACED by the fixture's (the constructor draws them from the
           global RNG, stream.jl:176-181: `empty!(W.sublist); append!(W.sublist, ...)` needs no patch of the reference), then the
           reference's own stream_new_line (stream.jl:625) for every (seed, offset) in the order of the driver loop
           (stream.jl:761-773, restated here in 8 lines because `stream` builds its StreamWork internally).
  LCM-guided tracking is not part of this: the reference samples with Julia's global RNG (stream.jl:452), the C ABI with a
  counter-based stream (include/fibers_hip.h): the two cannot agree line by line, by construction.
=#

length(ARGS) == 3 || error("usage: julia --threads 1 make_reference_fixtures.jl <Fibers.jl checkout> <raw input dir> <output dir>")
const FIBERS_DIR, RAW_DIR, OUT_DIR = ARGS

import Pkg
Pkg.activate(FIBERS_DIR)
using Fibers
const F = Fibers

# ---- exchange format ---------------------------------------------------------------------------------------------------------
const DTYPES = Dict("float32"=>Float32, "float64"=>Float64, "int32"=>Int32, "int64"=>Int64, "uint8"=>UInt8)
const DNAMES = Dict(v=>k for (k, v) in DTYPES)

function read_case(dir::String)
  out = Dict{String,Any}()
  for ln in eachline(joinpath(dir, "meta.txt"))
    ln = strip(ln)
    isempty(ln) && continue
    if occursin(" = ", ln)
      k, v = split(ln, " = ", limit=2)
      num = tryparse(Float64, v)
      out[String(k)] = isnothing(num) ? String(v) : num
      continue
    end
    p = split(ln)
    k, T, nd = String(p[1]), DTYPES[p[2]], parse(Int, p[3])
    dims = Tuple(parse.(Int, p[4:3+nd]))
    a = Array{T}(undef, dims...)
    open(io -> read!(io, a), joinpath(dir, k * ".bin"))
    out[k] = a
  end
  return out
end

function write_case(dir::String, arrays, scalars=Pair{String,Any}[])
  mkpath(dir)
  open(joinpath(dir, "meta.txt"), "w") do meta
    for (k, a0) in arrays
      a = a0 isa BitArray ? UInt8.(a0) : (a0 isa AbstractArray ? Array(a0) : [a0])
      println(meta, k, " ", DNAMES[eltype(a)], " ", ndims(a), " ", join(size(a), " "))
      open(io -> write(io, a), joinpath(dir, k * ".bin"), "w")
    end
    for (k, v) in scalars
      println(meta, k, " = ", v)
    end
  end
end

# ---- an MRI around an array: MRI(vol) leaves the header empty (mri.jl:138), the fits read volsize / nframes / volres ------------
function as_mri(vol::Array{T}; bval=nothing, bvec=nothing, volres=Float32[1, 1, 1]) where T<:Number
  m = F.MRI(vol)
  m.volsize = Int32[size(vol, 1), size(vol, 2), size(vol, 3)]
  m.height, m.width, m.depth = Int32(size(vol, 1)), Int32(size(vol, 2)), Int32(size(vol, 3))
  m.nframes = Int32(ndims(vol) > 3 ? size(vol, 4) : 1)
  m.nvoxels = Int32(prod(m.volsize))
  m.volres = Float32.(volres)
  m.xsize, m.ysize, m.zsize = m.volres
  isnothing(bval) || (m.bval = Float32.(vec(bval)))
  isnothing(bvec) || (m.bvec = Float32.(bvec))          # [nvol x 3], unit-normalised by the fixture generator like mri_read does (mri.jl:711-712)
  return m
end

vol3(m) = ndims(m.vol) == 4 ? m.vol[:, :, :, 1] : m.vol

sphere_of(name::AbstractString) = name == "sphere_362" ? F.sphere_362 : (name == "sphere_724" ? F.sphere_724 : F.sphere_642)

# ---- the fits ------------------------------------------------------------------------------------------------------------------
function run_dti(c)
  dwi  = as_mri(c["dwi"]; bval=c["bval"], bvec=c["bvec"])
  mask = as_mri(c["mask"])
  d = F.dti_fit(dwi, mask)
  adc, s0 = F.adc_fit(dwi, mask)
  return Pair{String,Any}["s0"=>vol3(d.s0), "eigval1"=>vol3(d.eigval1), "eigval2"=>vol3(d.eigval2), "eigval3"=>vol3(d.eigval3),
                          "eigvec1"=>d.eigvec1.vol, "eigvec2"=>d.eigvec2.vol, "eigvec3"=>d.eigvec3.vol,
                          "rd"=>vol3(d.rd), "md"=>vol3(d.md), "fa"=>vol3(d.fa), "adc"=>vol3(adc), "adc_s0"=>vol3(s0)]
end

function odf_outputs(r)
  out = Pair{String,Any}["odf"=>r.odf.vol]
  for k in 1:3
    push!(out, "peak$k"=>r.peak[k].vol)
    push!(out, "qa$k"=>vol3(r.qa[k]))
  end
  return out
end

function run_gqi(c)
  dwi  = as_mri(c["dwi"]; bval=c["bval"], bvec=c["bvec"])
  mask = as_mri(c["mask"])
  return odf_outputs(F.gqi_rec(dwi, mask, sphere_of(c["sphere"]), Float32(c["sigma"])))
end

function run_dsi(c)
  dwi  = as_mri(c["dwi"]; bval=c["bval"], bvec=c["bvec"])
  mask = as_mri(c["mask"])
  r = F.dsi_rec(dwi, mask, F.sphere_642, Int(c["hann_width"]))
  out = odf_outputs(r)
  pushfirst!(out, "pdf"=>r.pdf.vol)
  return out
end

function run_peaks(c)
  odf = c["odf"]                                          # [nvert x ncase]
  nvert, n = size(odf)
  W = F.GQIwork(Float32[0, 1000], Float32[0 0 0; 1 0 0], F.sphere_642)     # any b-table: only faces / o / odf_peak / isort are used
  top = Matrix{Int32}(undef, 3, n)
  nvalid = Vector{Int32}(undef, n)
  for v in 1:n
    W.o[1] .= odf[:, v]
    nvalid[v] = F.find_peaks!(W)                          # gqi.jl:180-201
    top[:, v] = Int32.(W.isort[1][1:3] .- 1)              # 0-based, like the C ABI
  end
  return Pair{String,Any}["isort_top"=>top, "nvalid"=>nvalid]
end

# ---- streamlines: the reference's StreamWork + stream_new_line, the fixture's offsets ---------------------------------------------
function trace_all(W, seed_mask::AbstractArray, sublist::Matrix{Float32})
  empty!(W.sublist)                                       # stream.jl:176-181 drew these from the global RNG: take the fixture's
  for i in 1:size(sublist, 1)
    push!(W.sublist, sublist[i, :])
  end
  npts = Int32[]
  xyz = Vector{Matrix{Float32}}()
  for vox in findall(seed_mask .> 0)                      # column-major order (stream.jl:744, :751)
    for isub in eachindex(W.sublist)                      # stream.jl:764-767
      strline, _ = F.stream_new_line(Int.([vox[1], vox[2], vox[3]]), W.sublist[isub], W)
      size(strline, 2) < W.len_min && continue            # stream.jl:769
      push!(npts, Int32(size(strline, 2)))
      push!(xyz, Float32.(strline))
    end
  end
  pts = isempty(xyz) ? zeros(Float32, 0, 3) : permutedims(reduce(hcat, xyz))     # [npoints x 3], 1-based voxel coordinates
  return npts, pts
end

function run_stream(c)
  ov = c["ovec"]                                          # [3 (vectors), nx, ny, nz, 3]
  nvec = size(ov, 1)
  ovs  = F.MRI[as_mri(ov[k, :, :, :, :]) for k in 1:nvec]
  fs   = F.MRI[as_mri(c["f"][k, :, :, :]) for k in 1:nvec]
  fa   = as_mri(c["fa"])
  mask = as_mri(c["mask"])
  kw = (f_thresh=c["kw_f_thresh"], fa_thresh=c["kw_fa_thresh"], len_min=Int(c["kw_len_min"]), ang_thresh=c["kw_ang_thresh"],
        step_size=c["kw_step_size"], smooth_coeff=c["kw_smooth_coeff"])
  nsub = size(c["sublist"], 1)
  W = F.StreamWork(ovs; f=fs, fa=fa, mask=mask, nsub=nsub, kw...)
  mn, mx = trace_all(W, c["seed"], c["sublist"])
  W1 = F.StreamWork(ovs[1]; mask=mask, nsub=nsub)         # defaults, seeds = the brain mask
  sn, sx = trace_all(W1, W1.mask, c["sublist"])
  return Pair{String,Any}["multi_npts"=>mn, "multi_xyz"=>mx, "single_npts"=>sn, "single_xyz"=>sx]
end

function run_micro(c)
  ov   = as_mri(c["ovec"]; volres=Float32[0.01, 0.01, 0.01])    # <= 50 um: the microscopy regime (stream.jl:85)
  f    = as_mri(c["f"]; volres=Float32[0.01, 0.01, 0.01])
  mask = as_mri(c["mask"]; volres=Float32[0.01, 0.01, 0.01])
  W = F.StreamWork(ov; f=f, f_thresh=c["kw_f_thresh"], mask=mask, nsub=0, len_max=Int(c["kw_len_max"]), ang_thresh=c["kw_ang_thresh"],
                   step_size=c["kw_step_size"], smooth_coeff=c["kw_smooth_coeff"], search_dist=Int(c["kw_search_dist"]),
                   search_ang=c["kw_search_ang"])
  n, x = trace_all(W, c["seed"], c["sublist"])
  return Pair{String,Any}["npts"=>n, "xyz"=>x]
end

const RUNNERS = Dict("dti"=>run_dti, "gqi"=>run_gqi, "dsi"=>run_dsi, "peaks"=>run_peaks, "stream"=>run_stream, "micro"=>run_micro)

for name in sort(readdir(RAW_DIR))
  dir = joinpath(RAW_DIR, name)
  isfile(joinpath(dir, "meta.txt")) || continue
  c = read_case(dir)
  kind = c["kind"]
  haskey(RUNNERS, kind) || continue
  println("reference: ", name, " (", kind, ")")
  out = RUNNERS[kind](c)
  write_case(joinpath(OUT_DIR, name), out,
             Pair{String,Any}["kind"=>kind, "source"=>"lincbrain/Fibers.jl, julia " * string(VERSION) * ", threads " * string(Threads.nthreads())])
end
println("done: ", OUT_DIR)
