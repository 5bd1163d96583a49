// setup_check.cpp -- csrc/setup.cpp's host-only table builders (pinv / SVD, DTI design, GQI matrix, DSI dense maps, face folding) under
// AddressSanitizer + UBSan (tests/test_host_sanitizers.py builds it with g++; no HIP call is made: setup.cpp's use_device and event
// timing are linked but never executed).  Inputs come from raw little-endian files the test writes with NumPy, outputs go back the same
// way and are compared with the oracle's NumPy restatements of DTIwork (dti.jl:101-155), GQIwork (gqi.jl:32-82) and the first rows of
// the DSI chain (dsi.jl:41-143) by the Python side.
// usage: setup_check <dir>   (reads dti_bval dti_bvec gqi_bval gqi_bvec dsi_bval dsi_bvec verts faces; writes dti_A dti_pA adc_pA gqi_A dsi_A dsi_meta nbr)
#include <cstdio>
#include <string>
#include <vector>

#include "../fibers.jl_amd/csrc/setup.cpp"

template <typename T>
static std::vector<T> rd(const std::string &dir, const char *name) {
    FILE *f = fopen((dir + "/" + name).c_str(), "rb");
    if (!f) { fprintf(stderr, "cannot read %s\n", name); exit(2); }
    fseek(f, 0, SEEK_END);
    const long n = ftell(f);
    fseek(f, 0, SEEK_SET);
    std::vector<T> v((size_t)n / sizeof(T));
    if (fread(v.data(), sizeof(T), v.size(), f) != v.size()) { fprintf(stderr, "short read of %s\n", name); exit(2); }
    fclose(f);
    return v;
}
template <typename T>
static void wr(const std::string &dir, const char *name, const std::vector<T> &v) {
    FILE *f = fopen((dir + "/" + name).c_str(), "wb");
    if (!f || fwrite(v.data(), sizeof(T), v.size(), f) != v.size()) { fprintf(stderr, "cannot write %s\n", name); exit(2); }
    fclose(f);
}

int main(int argc, char **argv) {
    if (argc < 2) return 2;
    const std::string dir = argv[1];
    const auto verts = rd<float>(dir, "verts"), dbval = rd<float>(dir, "dti_bval"), dbvec = rd<float>(dir, "dti_bvec"), gbval = rd<float>(dir, "gqi_bval"),
               gbvec = rd<float>(dir, "gqi_bvec"), sbval = rd<float>(dir, "dsi_bval"), sbvec = rd<float>(dir, "dsi_bvec");
    const auto faces = rd<int32_t>(dir, "faces");
    const int nverts = (int)verts.size() / 3, nvert = nverts / 2, nfaces = (int)faces.size() / 3;
    {   // DTIwork / ADCwork: design matrix + pinv (dti.jl:117-143, 68-72)
        const int n = (int)dbval.size();
        std::vector<float> A((size_t)n * 7), pA((size_t)7 * n), A2((size_t)n * 2), pA2((size_t)2 * n);
        fib::host_dti_design(dbval.data(), dbvec.data(), n, 7, A.data());
        fib::host_pinv(A.data(), n, 7, pA.data());
        fib::host_dti_design(dbval.data(), nullptr, n, 2, A2.data());
        fib::host_pinv(A2.data(), n, 2, pA2.data());
        wr(dir, "dti_A", A); wr(dir, "dti_pA", pA); wr(dir, "adc_pA", pA2);
        // a rank-deficient design (every gradient along x): the singular values below eps * max drop out, nothing divides by zero
        std::vector<float> bv((size_t)3 * n, 0.0f);
        for (int i = 0; i < n; i++) bv[i] = 1.0f;
        fib::host_dti_design(dbval.data(), bv.data(), n, 7, A.data());
        fib::host_pinv(A.data(), n, 7, pA.data());
        for (float x : pA) if (!(x == x) || x > 1e30f || x < -1e30f) { fprintf(stderr, "pinv of a rank-deficient design is not finite\n"); return 1; }
    }
    {   // GQIwork (gqi.jl:68-69) + the folded neighbour table (gqi.jl:63-64, 185-196)
        const int n = (int)gbval.size();
        std::vector<float> A((size_t)nvert * n);
        fib::host_gqi_matrix(gbval.data(), gbvec.data(), n, verts.data(), nverts, 1.25f, A.data());
        wr(dir, "gqi_A", A);
        std::vector<int32_t> nbr;
        int md = 0;
        if (fib::host_neighbours(faces.data(), nfaces, nverts, nbr, &md) != FIB_OK) { fprintf(stderr, "host_neighbours: %s\n", fib::last_error()); return 1; }
        nbr.push_back(md);
        wr(dir, "nbr", nbr);
        std::vector<int32_t> bad(faces);
        bad[3] = nverts + 5;                                   // a face that names a vertex outside the tessellation: an error code, no out-of-bounds write
        if (fib::host_neighbours(bad.data(), nfaces, nverts, nbr, &md) != FIB_ERR_INVALID) { fprintf(stderr, "bad face index accepted\n"); return 1; }
    }
    {   // DSIwork as two dense maps (dsi.jl:59-143 + the linear chain :204-242)
        const int n = (int)sbval.size();
        std::vector<float> A((size_t)(n + nvert) * n);
        int sf = -2; float sc = 0.0f;
        std::vector<int> iq;
        if (fib::host_dsi_matrix(sbval.data(), sbvec.data(), n, verts.data(), nverts, 32, A.data(), &sf, &sc, &iq) != FIB_OK) { fprintf(stderr, "host_dsi_matrix: %s\n", fib::last_error()); return 1; }
        wr(dir, "dsi_A", A);
        std::vector<float> meta = {(float)sf, sc};
        for (int v : iq) meta.push_back((float)v);
        wr(dir, "dsi_meta", meta);
        // one b-value only: the reference's minimum(bval[bval .> bmin]) throws; here an error code
        std::vector<float> flat((size_t)n, 1000.0f);
        if (fib::host_dsi_matrix(flat.data(), sbvec.data(), n, verts.data(), nverts, 32, A.data(), &sf, &sc, nullptr) != FIB_ERR_UNSUPPORTED) { fprintf(stderr, "flat b-table accepted\n"); return 1; }
    }
    printf("setup_check ok\n");
    return 0;
}
