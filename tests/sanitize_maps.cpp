// CPU sanitizer harness for the host-side index maps (dl_esm_inf_amd/csrc/dlesm_maps.cpp):
// built by tests/test_sanitizers.py with g++ -fsanitize=address,undefined and run over a sweep
// of decompositions, message tables, bounds and alignment strings.  (GPU AddressSanitizer is not
// available on the pool; the reference's own practice is valgrind + -fcheck=all on the CPU,
// example/Makefile:66,71 and compiler_setup/gnu.sh:7-8.)
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <unistd.h>

#include "dlesm_error.h"

namespace dlesm {
static char g_err[512];
int fail(int code, const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
    return code;
}
void clear_error() { g_err[0] = 0; }
} // namespace dlesm

static long checks = 0;
#define REQUIRE(c)                                                                     \
    do {                                                                               \
        checks++;                                                                      \
        if (!(c)) {                                                                    \
            fprintf(stderr, "FAILED %s (line %d): %s\n", #c, __LINE__, dlesm::g_err);  \
            return 1;                                                                  \
        }                                                                              \
    } while (0)

int main()
{
    // decompositions + message tables for every rank: each send must have its receive
    for (int n = 1; n <= 24; n++)
        for (int nx : {1, 2, 7, 16, 33, 100, 1000})
            for (int ny : {1, 3, 10, 32, 257}) {
                std::vector<dlesm_subdomain> subs(n);
                dlesm_decomp d;
                REQUIRE(dlesm_decompose(nx, ny, n, 0, 0, 1, &d, subs.data()) == 0);
                REQUIRE(d.nx * d.ny == n);
                long cells = 0;
                for (auto &s : subs) cells += (long)s.internal.nx * s.internal.ny;
                REQUIRE(cells == (long)nx * ny);
                bool empty = false;
                for (auto &s : subs) empty = empty || s.internal.nx < 1 || s.internal.ny < 1;
                std::vector<dlesm_comm_tables> t(n);
                for (int r = 1; r <= n; r++) {
                    int rc = dlesm_map_comms(&d, subs.data(), n, r, &t[r - 1]);
                    REQUIRE(empty ? rc == DLESM_EINVAL : rc == 0);
                }
                if (empty) continue;
                for (int r = 0; r < n; r++)
                    for (int k = 0; k < t[r].nsend; k++) {
                        const dlesm_comm_tables &p = t[t[r].destination[k]];
                        int hits = 0;
                        for (int q = 0; q < p.nrecv; q++)
                            hits += p.source[q] == r && p.dirrecv[q] == t[r].dirsend[k] &&
                                    p.nxrecv[q] == t[r].nxsend[k] && p.nyrecv[q] == t[r].nysend[k] &&
                                    p.idesrecv[q] == t[r].idessend[k] && p.jdesrecv[q] == t[r].jdessend[k];
                        REQUIRE(hits == 1);
                    }
                for (int ia = -1; ia <= nx + 1; ia += (nx > 40 ? 13 : 1))
                    for (int ja = -1; ja <= ny + 1; ja += (ny > 40 ? 11 : 1)) {
                        int o = dlesm_iprocmap(&d, subs.data(), n, ia, ja);
                        bool in = ia >= 1 && ia <= nx && ja >= 1 && ja <= ny;
                        REQUIRE(in ? (o >= 1 && o <= n) : o == 0);
                    }
            }
    // depth-d tables on halo_width-d decompositions: pairing, patch bounds, and "a halo cell is
    // written by exactly one message" over the whole depth-d frame that lies inside the domain
    for (int n : {1, 2, 3, 4, 6, 8, 9, 12})
        for (int depth : {1, 2, 3, 4, 6, 8})
            for (int nx : {16, 33, 100})
                for (int ny : {12, 40, 257}) {
                    std::vector<dlesm_subdomain> subs(n);
                    dlesm_decomp d;
                    REQUIRE(dlesm_decompose(nx, ny, n, 0, 0, depth, &d, subs.data()) == 0);
                    bool small = false;
                    for (auto &s : subs) small = small || s.internal.nx < depth || s.internal.ny < depth;
                    std::vector<dlesm_comm_tables> t(n);
                    for (int r = 1; r <= n; r++) {
                        int rc = dlesm_map_comms_depth(&d, subs.data(), n, r, depth, &t[r - 1]);
                        REQUIRE(small ? rc == DLESM_EINVAL : rc == 0);
                    }
                    if (small) continue;
                    for (int r = 0; r < n; r++) {
                        const dlesm_region &in = subs[r].internal;
                        const int W = in.nx + 2 * depth, Hh = in.ny + 2 * depth;
                        std::vector<int> hits((size_t)W * Hh, 0);
                        for (int k = 0; k < t[r].nrecv; k++) {
                            for (int j = 0; j < t[r].nyrecv[k]; j++)
                                for (int i = 0; i < t[r].nxrecv[k]; i++) {
                                    const int ii = t[r].idesrecv[k] + i, jj = t[r].jdesrecv[k] + j;
                                    REQUIRE(ii >= 1 && ii <= W && jj >= 1 && jj <= Hh);
                                    hits[(size_t)(jj - 1) * W + ii - 1]++;
                                }
                            const dlesm_comm_tables &p = t[t[r].source[k]];
                            int m = 0;
                            for (int q = 0; q < p.nsend; q++)
                                m += p.destination[q] == r && p.dirsend[q] == t[r].dirrecv[k] &&
                                     p.nxsend[q] == t[r].nxrecv[k] && p.nysend[q] == t[r].nyrecv[k] &&
                                     p.idessend[q] == t[r].idesrecv[k] && p.jdessend[q] == t[r].jdesrecv[k];
                            REQUIRE(m == 1);
                        }
                        for (int jj = 1; jj <= Hh; jj++)
                            for (int ii = 1; ii <= W; ii++) {
                                const int gi = subs[r].global.xstart + ii - in.xstart;
                                const int gj = subs[r].global.ystart + jj - in.ystart;
                                const bool internal = ii >= in.xstart && ii <= in.xstop && jj >= in.ystart && jj <= in.ystop;
                                const bool in_domain = gi >= 1 && gi <= nx && gj >= 1 && gj <= ny;
                                const int h = hits[(size_t)(jj - 1) * W + ii - 1];
                                if (internal) REQUIRE(h == 0);
                                else if (in_domain) REQUIRE(h == 1);
                                else REQUIRE(h <= 1);      // ring cells carried along the strips
                            }
                    }
                }
    {
        std::vector<dlesm_subdomain> subs(4);
        dlesm_decomp d;
        dlesm_comm_tables t;
        REQUIRE(dlesm_decompose(40, 40, 4, 0, 0, 2, &d, subs.data()) == 0);
        REQUIRE(dlesm_map_comms_depth(&d, subs.data(), 4, 1, 3, &t) == DLESM_EINVAL);   // halo width 2 < depth 3
        REQUIRE(dlesm_map_comms_depth(&d, subs.data(), 4, 1, 0, &t) == DLESM_EINVAL);
        REQUIRE(dlesm_map_comms_depth(&d, subs.data(), 4, 5, 2, &t) == DLESM_EINVAL);
        REQUIRE(dlesm_map_comms_depth(&d, subs.data(), 3, 1, 2, &t) == DLESM_EINVAL);
        REQUIRE(dlesm_map_comms_depth(nullptr, subs.data(), 4, 1, 2, &t) == DLESM_EINVAL);
    }
    // user tilings and rejected arguments
    {
        std::vector<dlesm_subdomain> subs(64);
        dlesm_decomp d;
        REQUIRE(dlesm_decompose(100, 80, 12, 3, 4, 1, &d, subs.data()) == 0 && d.nx == 3 && d.ny == 4);
        REQUIRE(dlesm_decompose(100, 80, 12, 3, 5, 1, &d, subs.data()) == DLESM_EINVAL);
        REQUIRE(dlesm_decompose(100, 80, 12, 3, 0, 1, &d, subs.data()) == DLESM_EABORT);
        REQUIRE(dlesm_decompose(100, 80, 0, 0, 0, 1, &d, subs.data()) == DLESM_EINVAL);
        REQUIRE(dlesm_decompose(100, 80, 4, 0, 0, 1, nullptr, subs.data()) == DLESM_EINVAL);
    }
    // bounds for every point type / offset / BC
    for (int pt = -1; pt <= 5; pt++)
        for (int off = 0; off <= 4; off++)
            for (int bx = 0; bx <= 2; bx++)
                for (int by = 0; by <= 2; by++) {
                    dlesm_region sub{10, 7, 2, 11, 2, 8}, in, wh;
                    int rc = dlesm_field_bounds(pt, off, bx, by, &sub, 14, 10, &in, &wh);
                    REQUIRE(rc == 0 || rc == DLESM_EABORT);
                    if (rc == 0) REQUIRE(wh.nx == in.nx + 2 && wh.ny == in.ny + 2 && in.nx >= 1);
                }
    // alignment strings
    const char *good[] = {"1", "8", "64", "999", " 16", "16 ", "+4"};
    const char *bad[] = {"0", "-8", "abc", "1024", "8x", "", "+", "- 1", "1e2"};
    int a;
    for (const char *g : good) {
        setenv("DL_ESM_ALIGNMENT", g, 1);
        REQUIRE(dlesm_alignment_from_env(&a) == 0 && a >= 1);
    }
    for (const char *b : bad) {
        setenv("DL_ESM_ALIGNMENT", b, 1);
        REQUIRE(dlesm_alignment_from_env(&a) == DLESM_EABORT);
    }
    unsetenv("DL_ESM_ALIGNMENT");
    REQUIRE(dlesm_alignment_from_env(&a) == 0 && a == 1);
    int nx, ny;
    for (int al : {0, 1, 2, 3, 8, 64, 999})
        for (int w : {1, 3, 63, 64, 65, 16386}) {
            REQUIRE(dlesm_grid_extents(w, 5, al, &nx, &ny) == 0);
            REQUIRE(nx > w && nx % (al > 0 ? al : 1) == 0 && nx - w <= (al > 0 ? al : 1) && ny == 6);
        }
    // periodic halo regions: sources inside the internal region (+ halo columns for the y pair),
    // destinations on the ring, extents equal
    for (int nx : {1, 2, 10, 257})
        for (int ny : {1, 3, 64})
            for (int bx : {0, 1})
                for (int by : {0, 1}) {
                    dlesm_region it{nx, ny, 2, nx + 1, 2, ny + 1}, src[4], dst[4];
                    int n = -1;
                    REQUIRE(dlesm_periodic_halos(&it, bx, by, src, dst, &n) == 0);
                    REQUIRE(n == 2 * ((bx == 0) + (by == 0)));
                    for (int k = 0; k < n; k++) {
                        REQUIRE(src[k].nx == dst[k].nx && src[k].ny == dst[k].ny && src[k].nx >= 1 && src[k].ny >= 1);
                        REQUIRE(dst[k].xstart >= 1 && dst[k].ystart >= 1 && dst[k].xstop <= nx + 2 && dst[k].ystop <= ny + 2);
                    }
                }
    // the id rendezvous: publish / fetch / stale records / acknowledgements
    {
        char path[256], id[DLESM_UNIQUE_ID_BYTES], got[DLESM_UNIQUE_ID_BYTES];
        snprintf(path, sizeof path, "/tmp/dlesm_sanitize_rv_%ld", (long)getpid());
        for (int k = 0; k < DLESM_UNIQUE_ID_BYTES; k++) id[k] = (char)(3 * k + 1);
        REQUIRE(dlesm_rendezvous_remove(path) == 0);
        REQUIRE(dlesm_rendezvous_fetch(path, got, "2:x", 30) == DLESM_EINVAL);
        REQUIRE(dlesm_rendezvous_publish(path, id, "2:x") == 0);
        REQUIRE(dlesm_rendezvous_fetch(path, got, "2:x", 30) == 0 && memcmp(id, got, sizeof id) == 0);
        REQUIRE(dlesm_rendezvous_fetch(path, got, "2:y", 30) == DLESM_EINVAL);
        char longtok[200];
        memset(longtok, 'a', sizeof longtok - 1);
        longtok[sizeof longtok - 1] = 0;
        REQUIRE(dlesm_rendezvous_publish(path, id, longtok) == DLESM_EINVAL);
        REQUIRE(dlesm_rendezvous_ack(path, 1) == 0);
        REQUIRE(dlesm_rendezvous_wait_acks(path, 2, 100) == 0);
        REQUIRE(dlesm_rendezvous_wait_acks(path, 3, 30) == DLESM_EINVAL);
        REQUIRE(dlesm_rendezvous_remove(path) == 0);
    }
    printf("sanitize_maps: %ld checks passed\n", checks);
    return 0;
}
