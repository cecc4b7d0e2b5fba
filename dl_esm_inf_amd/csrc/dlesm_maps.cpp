// Host-side index maps of the dl_esm_inf hot path: array extents, per-field
// iteration bounds, the 2-D block decomposition and the halo-exchange message
// tables.  Pure integer work, no device involved.
//
// The reference derives the message tables by scanning sub-domain borders with
// an owner search per point (parallel_comms_mod.f90:296-1170), an algorithm
// written for irregular partitions.  go_decompose only ever produces a
// tensor-product mesh of tiles (parallel_mod.f90:244-317), so here the mesh is
// recovered once and every message is written down directly from the tile's
// mesh coordinates -- O(1) per rank instead of O(perimeter * nranks) -- in the
// same order and with the same values as the reference's tables.
#include <cmath>
#include <cstdlib>
#include <cstring>

#include "dlesm_error.h"   // no HIP here: this file also builds stand-alone (tests/sanitize_maps.cpp)

using dlesm::fail;

extern "C" int dlesm_alignment_from_env(int *alignment)
{
    // grid_mod.f90:344-363: character(len=3) buffer, read with "(i3)"
    if (!alignment) return fail(DLESM_EINVAL, "alignment pointer is null");
    const char *v = std::getenv("DL_ESM_ALIGNMENT");
    if (!v) { // status 1: not present -> no padding beyond the mandatory +1
        *alignment = 1;
        return DLESM_OK;
    }
    size_t len = std::strlen(v);
    while (len > 0 && v[len - 1] == ' ') len--; // trailing blanks do not count as truncation
    if (len > 3)
        return fail(DLESM_EABORT, "Error: Only numbers of up to 3 digits are supported in the "
                                  "DL_ESM_ALIGNMENT environment variable.");
    // an I3 edit descriptor ignores blanks and accepts an optional sign
    long val = 0;
    int sign = 1, ndig = 0;
    bool bad = false, seen_sign = false;
    for (size_t k = 0; k < len; k++) {
        char ch = v[k];
        if (ch == ' ') continue;
        if ((ch == '+' || ch == '-') && ndig == 0 && !seen_sign) {
            seen_sign = true;
            sign = ch == '-' ? -1 : 1;
        } else if (ch >= '0' && ch <= '9') {
            val = val * 10 + (ch - '0');
            ndig++;
        } else {
            bad = true;
        }
    }
    val *= sign;
    if (bad || (seen_sign && ndig == 0) || val < 1)
        return fail(DLESM_EABORT, "Error: Cannot convert DL_ESM_ALIGNMENT value (%.3s) into a "
                                  "positive integer.", v);
    *alignment = (int)val;
    return DLESM_OK;
}

extern "C" int dlesm_grid_extents(int sub_global_nx, int sub_global_ny, int alignment, int *nx,
                                  int *ny)
{
    if (!nx || !ny) return fail(DLESM_EINVAL, "null output pointer");
    if (sub_global_nx < 1 || sub_global_ny < 1)
        return fail(DLESM_EINVAL, "subdomain extent %dx%d", sub_global_nx, sub_global_ny);
    const int a = alignment > 0 ? alignment : 1;
    // at least one extra column for the staggered points, then round up to a
    // multiple of the alignment (grid_mod.f90:364-369); one extra row (grid:385)
    *nx = (sub_global_nx / a + 1) * a;
    *ny = sub_global_ny + 1;
    return DLESM_OK;
}

namespace {
void finish_region(dlesm_region &r)
{
    r.nx = r.xstop - r.xstart + 1;
    r.ny = r.ystop - r.ystart + 1;
}
} // namespace

extern "C" int dlesm_field_bounds(int grid_points, int offset, int bc_x, int bc_y,
                                  const dlesm_region *sub, int grid_nx, int grid_ny,
                                  dlesm_region *internal, dlesm_region *whole)
{
    if (!sub || !internal || !whole) return fail(DLESM_EINVAL, "null region pointer");
    const bool per_x = bc_x == DLESM_BC_PERIODIC, per_y = bc_y == DLESM_BC_PERIODIC;
    dlesm_region in = *sub;

    if (grid_points == DLESM_ALL_POINTS) {
        // a field on every point of the grid (field_mod.f90:628-648)
        in.xstart = 1; in.xstop = grid_nx;
        in.ystart = 1; in.ystop = grid_ny;
    } else if (grid_points < DLESM_U_POINTS || grid_points > DLESM_ALL_POINTS) {
        return fail(DLESM_EABORT, "r2d_field_constructor: ERROR: invalid specifier for type of "
                                  "mesh points");
    } else if (offset == DLESM_OFFSET_NE) {
        // NE staggering: all four point types iterate over the T-point internal
        // region; periodic boundaries are not implemented by the reference
        // (field_mod.f90:762-781, 876-893, 979-998, 1104-1120)
        static const char *who[] = {"cu_ne_init", "cv_ne_init", "ct_ne_init", "cf_ne_init"};
        if (per_x || per_y)
            return fail(DLESM_EABORT, "ERROR: %s: implement periodic BCs!", who[grid_points]);
    } else if (offset == DLESM_OFFSET_SW) {
        // SW staggering (field_mod.f90:675-751, 813-868, 922-961, 1027-1084)
        if (grid_points == DLESM_U_POINTS && !per_x) in.xstart = sub->xstart + 1;
        if (grid_points == DLESM_V_POINTS && !per_y)
            return fail(DLESM_EABORT, "cv_sw_init: IMPLEMENT non-periodic BCs!");
        if (grid_points == DLESM_F_POINTS && (!per_x || !per_y))
            return fail(DLESM_EABORT, "cf_sw_init: CHECK non-periodic BCs!");
    } else {
        return fail(DLESM_EABORT, "field_init: ERROR - unsupported grid offset!");
    }
    finish_region(in);
    // NBOUNDARY = 1 ring on every side, whatever the BC (field_mod.f90:227,606-622)
    dlesm_region wh = in;
    wh.xstart -= 1; wh.xstop += 1;
    wh.ystart -= 1; wh.ystop += 1;
    finish_region(wh);
    *internal = in;
    *whole = wh;
    return DLESM_OK;
}

// init_periodic_bc_halos, field_mod.f90:1394-1464: for each periodic direction two copies,
// (source, destination) regions, in the reference's order: x first -- east halo column <- west-most
// internal column, west halo column <- east-most internal column, over the internal rows -- then y
// -- north halo row <- south-most internal row, south halo row <- north-most internal row, over the
// internal columns PLUS the two halo columns, so that the corners come out right when the copies
// are applied in this order.
extern "C" int dlesm_periodic_halos(const dlesm_region *it, int bc_x, int bc_y, dlesm_region *source,
                                    dlesm_region *dest, int *num_halos)
{
    if (!it || !source || !dest || !num_halos) return fail(DLESM_EINVAL, "null pointer");
    int n = 0;
    auto add = [&](int sx0, int sx1, int sy0, int sy1, int dx0, int dx1, int dy0, int dy1) {
        source[n] = dlesm_region{sx1 - sx0 + 1, sy1 - sy0 + 1, sx0, sx1, sy0, sy1};
        dest[n] = dlesm_region{dx1 - dx0 + 1, dy1 - dy0 + 1, dx0, dx1, dy0, dy1};
        n++;
    };
    if (bc_x == DLESM_BC_PERIODIC) {
        add(it->xstart, it->xstart, it->ystart, it->ystop, it->xstop + 1, it->xstop + 1, it->ystart, it->ystop);
        add(it->xstop, it->xstop, it->ystart, it->ystop, it->xstart - 1, it->xstart - 1, it->ystart, it->ystop);
    }
    if (bc_y == DLESM_BC_PERIODIC) {
        add(it->xstart - 1, it->xstop + 1, it->ystart, it->ystart, it->xstart - 1, it->xstop + 1, it->ystop + 1, it->ystop + 1);
        add(it->xstart - 1, it->xstop + 1, it->ystop, it->ystop, it->xstart - 1, it->xstop + 1, it->ystart - 1, it->ystart - 1);
    }
    *num_halos = n;
    return DLESM_OK;
}

extern "C" int dlesm_decompose(int domainx, int domainy, int ndomains, int ntilex, int ntiley,
                               int halo_width, dlesm_decomp *d, dlesm_subdomain *subs)
{
    if (!d || !subs) return fail(DLESM_EINVAL, "null output pointer");
    if (domainx < 1 || domainy < 1 || ndomains < 1)
        return fail(DLESM_EINVAL, "go_decompose: bad arguments %dx%d / %d", domainx, domainy,
                    ndomains);
    if (halo_width < 0) return fail(DLESM_EINVAL, "negative halo width");
    int P, Q;
    if (ntilex > 0 && ntiley > 0) {
        if (ntilex * ntiley != ndomains)
            return fail(DLESM_EINVAL, "go_decompose: %d x %d tiles != %d domains", ntilex, ntiley,
                        ndomains);
        P = ntilex;
        Q = ntiley;
    } else if (ntilex <= 0 && ntiley <= 0) {
        // most-square factorisation with the longer mesh side along the longer
        // domain side (parallel_mod.f90:167-190); single-precision sqrt as there
        int small = (int)std::sqrt((float)ndomains);
        while (ndomains % small) small--;
        int large = ndomains / small;
        if (domainx > domainy) { P = large; Q = small; }
        else { P = small; Q = large; }
    } else {
        return fail(DLESM_EABORT, "go_decompose: invalid arguments supplied");
    }
    d->global_nx = domainx;
    d->global_ny = domainy;
    d->nx = P;
    d->ny = Q;
    d->ndomains = ndomains;
    d->max_width = d->max_height = 0;

    // the first (X mod P) columns / (Y mod Q) rows are one cell wider (pmod:204-270)
    const int w0 = domainx / P, wrem = domainx % P;
    const int h0 = domainy / Q, hrem = domainy % Q;
    for (int iy = 0; iy < Q; iy++) {
        const int h = h0 + (iy < hrem ? 1 : 0);
        const int gy0 = iy * h0 + (iy < hrem ? iy : hrem) + 1;
        for (int ix = 0; ix < P; ix++) {
            const int w = w0 + (ix < wrem ? 1 : 0);
            const int gx0 = ix * w0 + (ix < wrem ? ix : wrem) + 1;
            dlesm_subdomain &s = subs[iy * P + ix]; // x fastest (pmod:244-317)
            s.internal.xstart = halo_width + 1;
            s.internal.xstop = halo_width + w;
            s.internal.ystart = halo_width + 1;
            s.internal.ystop = halo_width + h;
            s.internal.nx = w;
            s.internal.ny = h;
            s.global.xstart = gx0;
            s.global.xstop = gx0 + w - 1;
            s.global.ystart = gy0;
            s.global.ystop = gy0 + h - 1;
            s.global.nx = w + 2 * halo_width; // WHOLE width incl. halos (pmod:281)
            s.global.ny = h + 2 * halo_width;
            if (s.global.nx > d->max_width) d->max_width = s.global.nx;
            if (s.global.ny > d->max_height) d->max_height = s.global.ny;
        }
    }
    return DLESM_OK;
}

extern "C" int dlesm_iprocmap(const dlesm_decomp *d, const dlesm_subdomain *subs, int nranks,
                              int ia, int ja)
{
    if (!d || !subs) return 0;
    // tensor-product mesh: find the column, then the row (the reference scans
    // all ranks, parallel_comms_mod.f90:1388-1396)
    const int P = d->nx, Q = d->ny;
    if (P * Q > nranks || P < 1 || Q < 1) {
        for (int r = 0; r < nranks; r++) {
            const dlesm_region &g = subs[r].global;
            if (g.xstart <= ia && ia <= g.xstop && g.ystart <= ja && ja <= g.ystop) return r + 1;
        }
        return 0;
    }
    int ix = -1, iy = -1;
    for (int k = 0; k < P; k++)
        if (subs[k].global.xstart <= ia && ia <= subs[k].global.xstop) { ix = k; break; }
    for (int k = 0; k < Q; k++)
        if (subs[k * P].global.ystart <= ja && ja <= subs[k * P].global.ystop) { iy = k; break; }
    return (ix < 0 || iy < 0) ? 0 : iy * P + ix + 1;
}

namespace {

struct TableWriter {
    dlesm_comm_tables *t;
    int err = 0;
    void send(int dir, int dest, int isrc, int jsrc, int ides, int jdes, int nx, int ny)
    {
        if (t->nsend >= DLESM_MAXCOMM) { err = DLESM_ECOMMS; return; }
        const int k = t->nsend++;
        t->dirsend[k] = dir; t->destination[k] = dest;
        t->isrcsend[k] = isrc; t->jsrcsend[k] = jsrc;
        t->idessend[k] = ides; t->jdessend[k] = jdes;
        t->nxsend[k] = nx; t->nysend[k] = ny;
    }
    void recv(int dir, int src, int isrc, int jsrc, int ides, int jdes, int nx, int ny)
    {
        if (t->nrecv >= DLESM_MAXCOMM) { err = DLESM_ECOMMS; return; }
        const int k = t->nrecv++;
        t->dirrecv[k] = dir; t->source[k] = src;
        t->isrcrecv[k] = isrc; t->jsrcrecv[k] = jsrc;
        t->idesrecv[k] = ides; t->jdesrecv[k] = jdes;
        t->nxrecv[k] = nx; t->nyrecv[k] = ny;
    }
};

// is the decomposition the tensor-product mesh go_decompose builds?
bool is_tile_mesh(const dlesm_decomp *d, const dlesm_subdomain *subs)
{
    const int P = d->nx, Q = d->ny;
    for (int iy = 0; iy < Q; iy++)
        for (int ix = 0; ix < P; ix++) {
            const dlesm_subdomain &s = subs[iy * P + ix];
            const dlesm_subdomain &col = subs[ix], &row = subs[iy * P];
            if (s.global.xstart != col.global.xstart || s.global.xstop != col.global.xstop ||
                s.global.ystart != row.global.ystart || s.global.ystop != row.global.ystop)
                return false;
            if (ix > 0 && s.global.xstart != subs[iy * P + ix - 1].global.xstop + 1) return false;
            if (iy > 0 && s.global.ystart != subs[(iy - 1) * P + ix].global.ystop + 1) return false;
            if (s.internal.nx < 1 || s.internal.ny < 1) return false;
            if (s.internal.xstop - s.internal.xstart + 1 != s.global.xstop - s.global.xstart + 1)
                return false;
            if (s.internal.ystop - s.internal.ystart + 1 != s.global.ystop - s.global.ystart + 1)
                return false;
        }
    return true;
}

} // namespace

extern "C" int dlesm_map_comms(const dlesm_decomp *d, const dlesm_subdomain *subs, int nranks,
                               int rank1, dlesm_comm_tables *t)
{
    if (!d || !subs || !t) return fail(DLESM_EINVAL, "null pointer");
    const int P = d->nx, Q = d->ny;
    if (P < 1 || Q < 1 || P * Q != d->ndomains || nranks != d->ndomains)
        return fail(DLESM_EINVAL, "map_comms: %d ranks for a %dx%d mesh of %d subdomains", nranks, P,
                    Q, d->ndomains);
    if (rank1 < 1 || rank1 > nranks) return fail(DLESM_EINVAL, "map_comms: rank %d of %d", rank1, nranks);
    if (!is_tile_mesh(d, subs))
        return fail(DLESM_EINVAL, "map_comms: decomposition is not a regular mesh of non-empty tiles");

    // unset slots carry the reference's sentinel (parallel_comms_mod.f90:246-261)
    for (int *p = &t->dirsend[0]; p < &t->nyrecv[0] + DLESM_MAXCOMM; p++) *p = -999;
    t->nsend = t->nrecv = 0;
    TableWriter w{t};

    const int r = rank1 - 1, ix = r % P, iy = r / P;
    const dlesm_region &me = subs[r].internal;
    const int wdt = me.nx, hgt = me.ny;
    const bool hasW = ix > 0, hasE = ix < P - 1, hasS = iy > 0, hasN = iy < Q - 1;
    // the receive halo sits at index 1 on the low side whatever the decomposition's
    // halo width (parallel_comms_mod.f90:334,524,700,868: "halo runs from 1..depth")
    const int lo_halo = 1;

    // ---- edges, in the reference's order: west, east, south, north.
    // A strip carries internal cells only; x-strips are 1 x height, y-strips width x 1.
    if (hasW) { // direction code Iplus: data needed by the west neighbour's (i+1) reads
        const dlesm_region &o = subs[r - 1].internal;
        w.send(DLESM_IPLUS, r - 1, me.xstart, me.ystart, o.xstop + 1, o.ystart, 1, hgt);
        w.recv(DLESM_IMINUS, r - 1, o.xstop, o.ystart, lo_halo, me.ystart, 1, hgt);
    }
    if (hasE) {
        const dlesm_region &o = subs[r + 1].internal;
        w.send(DLESM_IMINUS, r + 1, me.xstop, me.ystart, lo_halo, o.ystart, 1, hgt);
        // isrcrecv is xstart+1 in the reference's table (pcomms:521); it is never
        // read by the exchange, we keep the value for table parity
        w.recv(DLESM_IPLUS, r + 1, o.xstart + 1, o.ystart, me.xstop + 1, me.ystart, 1, hgt);
    }
    if (hasS) {
        const dlesm_region &o = subs[r - P].internal;
        w.send(DLESM_JPLUS, r - P, me.xstart, me.ystart, o.xstart, o.ystop + 1, wdt, 1);
        w.recv(DLESM_JMINUS, r - P, o.xstart, o.ystop, me.xstart, lo_halo, wdt, 1);
    }
    if (hasN) {
        const dlesm_region &o = subs[r + P].internal;
        w.send(DLESM_JMINUS, r + P, me.xstart, me.ystop, o.xstart, lo_halo, wdt, 1);
        w.recv(DLESM_JPLUS, r + P, o.xstart, o.ystart, me.xstart, me.ystop + 1, wdt, 1);
    }
    // ---- corners: a single cell to the diagonal tile, only where both edge
    // neighbours exist (pcomms:1039-1040); order SW, NE, NW, SE = codes 5..8;
    // the matching receive is filed under the opposite code (pcomms:1156-1163)
    if (hasW && hasS) {
        const dlesm_region &o = subs[r - P - 1].internal;
        w.send(DLESM_IPLUSJPLUS, r - P - 1, me.xstart, me.ystart, o.xstop + 1, o.ystop + 1, 1, 1);
        w.recv(DLESM_IMINUSJMINUS, r - P - 1, o.xstop, o.ystop, me.xstart - 1, me.ystart - 1, 1, 1);
    }
    if (hasE && hasN) {
        const dlesm_region &o = subs[r + P + 1].internal;
        w.send(DLESM_IMINUSJMINUS, r + P + 1, me.xstop, me.ystop, o.xstart - 1, o.ystart - 1, 1, 1);
        w.recv(DLESM_IPLUSJPLUS, r + P + 1, o.xstart, o.ystart, me.xstop + 1, me.ystop + 1, 1, 1);
    }
    if (hasW && hasN) {
        const dlesm_region &o = subs[r + P - 1].internal;
        w.send(DLESM_IPLUSJMINUS, r + P - 1, me.xstart, me.ystop, o.xstop + 1, o.ystart - 1, 1, 1);
        w.recv(DLESM_IMINUSJPLUS, r + P - 1, o.xstop, o.ystart, me.xstart - 1, me.ystop + 1, 1, 1);
    }
    if (hasE && hasS) {
        const dlesm_region &o = subs[r - P + 1].internal;
        w.send(DLESM_IMINUSJPLUS, r - P + 1, me.xstop, me.ystart, o.xstart - 1, o.ystop + 1, 1, 1);
        w.recv(DLESM_IPLUSJMINUS, r - P + 1, o.xstart, o.ystop, me.xstop + 1, me.ystart - 1, 1, 1);
    }
    if (w.err) return fail(w.err, "ERROR: Number of separate communications exceeds maximum of %d",
                           DLESM_MAXCOMM);
    return DLESM_OK;
}

// Depth-d tables (an extension: the reference stops at MAX_HALO_DEPTH = 1,
// parallel_comms_mod.f90:48,220-222).  Same neighbours, directions and order as above;
// strips are d cells deep and sit directly against the internal region, corners are d x d.
// Needs a decomposition whose halo width is >= d and tiles at least d cells wide and high.
extern "C" int dlesm_map_comms_depth(const dlesm_decomp *d, const dlesm_subdomain *subs, int nranks,
                                     int rank1, int depth, dlesm_comm_tables *t)
{
    if (!d || !subs || !t) return fail(DLESM_EINVAL, "null pointer");
    const int P = d->nx, Q = d->ny;
    if (P < 1 || Q < 1 || P * Q != d->ndomains || nranks != d->ndomains)
        return fail(DLESM_EINVAL, "map_comms_depth: %d ranks for a %dx%d mesh of %d subdomains", nranks, P,
                    Q, d->ndomains);
    if (rank1 < 1 || rank1 > nranks) return fail(DLESM_EINVAL, "map_comms_depth: rank %d of %d", rank1, nranks);
    if (depth < 1) return fail(DLESM_EINVAL, "map_comms_depth: depth %d", depth);
    if (!is_tile_mesh(d, subs))
        return fail(DLESM_EINVAL, "map_comms_depth: decomposition is not a regular mesh of non-empty tiles");
    for (int k = 0; k < nranks; k++) {
        const dlesm_region &in = subs[k].internal;
        if (in.xstart - 1 < depth || in.ystart - 1 < depth)
            return fail(DLESM_EINVAL, "map_comms_depth: halo width %d of the decomposition is less than depth %d",
                        (in.xstart < in.ystart ? in.xstart : in.ystart) - 1, depth);
        if (in.nx < depth || in.ny < depth)
            return fail(DLESM_EINVAL, "map_comms_depth: tile %d is %dx%d, smaller than depth %d", k, in.nx, in.ny,
                        depth);
    }
    for (int *p = &t->dirsend[0]; p < &t->nyrecv[0] + DLESM_MAXCOMM; p++) *p = -999;
    t->nsend = t->nrecv = 0;
    TableWriter w{t};
    const int r = rank1 - 1, ix = r % P, iy = r / P, n = depth;
    const dlesm_region &me = subs[r].internal;
    const int wdt = me.nx, hgt = me.ny;
    const bool hasW = ix > 0, hasE = ix < P - 1, hasS = iy > 0, hasN = iy < Q - 1;
    // low/high strips of a tile's internal region and of its halo, as first index
    auto lo_in = [](int start) { return start; };
    auto hi_in = [n](int stop) { return stop - n + 1; };
    auto lo_halo = [n](int start) { return start - n; };
    auto hi_halo = [](int stop) { return stop + 1; };
    // Where a strip ends at the edge of the domain it also carries the boundary-ring cell next to
    // it: a stage box grown into the halo reads the ring cells that lie under the NEIGHBOUR's
    // tile, and the neighbour is the one that holds them (in its own whole region).
    const int bS = hasS ? 0 : 1, bN = hasN ? 0 : 1, bW = hasW ? 0 : 1, bE = hasE ? 0 : 1;
    if (hasW) {
        const dlesm_region &o = subs[r - 1].internal;
        w.send(DLESM_IPLUS, r - 1, lo_in(me.xstart), me.ystart - bS, hi_halo(o.xstop), o.ystart - bS, n, hgt + bS + bN);
        w.recv(DLESM_IMINUS, r - 1, hi_in(o.xstop), o.ystart - bS, lo_halo(me.xstart), me.ystart - bS, n, hgt + bS + bN);
    }
    if (hasE) {
        const dlesm_region &o = subs[r + 1].internal;
        w.send(DLESM_IMINUS, r + 1, hi_in(me.xstop), me.ystart - bS, lo_halo(o.xstart), o.ystart - bS, n, hgt + bS + bN);
        w.recv(DLESM_IPLUS, r + 1, lo_in(o.xstart), o.ystart - bS, hi_halo(me.xstop), me.ystart - bS, n, hgt + bS + bN);
    }
    if (hasS) {
        const dlesm_region &o = subs[r - P].internal;
        w.send(DLESM_JPLUS, r - P, me.xstart - bW, lo_in(me.ystart), o.xstart - bW, hi_halo(o.ystop), wdt + bW + bE, n);
        w.recv(DLESM_JMINUS, r - P, o.xstart - bW, hi_in(o.ystop), me.xstart - bW, lo_halo(me.ystart), wdt + bW + bE, n);
    }
    if (hasN) {
        const dlesm_region &o = subs[r + P].internal;
        w.send(DLESM_JMINUS, r + P, me.xstart - bW, hi_in(me.ystop), o.xstart - bW, lo_halo(o.ystart), wdt + bW + bE, n);
        w.recv(DLESM_JPLUS, r + P, o.xstart - bW, lo_in(o.ystart), me.xstart - bW, hi_halo(me.ystop), wdt + bW + bE, n);
    }
    if (hasW && hasS) {
        const dlesm_region &o = subs[r - P - 1].internal;
        w.send(DLESM_IPLUSJPLUS, r - P - 1, lo_in(me.xstart), lo_in(me.ystart), hi_halo(o.xstop), hi_halo(o.ystop), n, n);
        w.recv(DLESM_IMINUSJMINUS, r - P - 1, hi_in(o.xstop), hi_in(o.ystop), lo_halo(me.xstart), lo_halo(me.ystart), n, n);
    }
    if (hasE && hasN) {
        const dlesm_region &o = subs[r + P + 1].internal;
        w.send(DLESM_IMINUSJMINUS, r + P + 1, hi_in(me.xstop), hi_in(me.ystop), lo_halo(o.xstart), lo_halo(o.ystart), n, n);
        w.recv(DLESM_IPLUSJPLUS, r + P + 1, lo_in(o.xstart), lo_in(o.ystart), hi_halo(me.xstop), hi_halo(me.ystop), n, n);
    }
    if (hasW && hasN) {
        const dlesm_region &o = subs[r + P - 1].internal;
        w.send(DLESM_IPLUSJMINUS, r + P - 1, lo_in(me.xstart), hi_in(me.ystop), hi_halo(o.xstop), lo_halo(o.ystart), n, n);
        w.recv(DLESM_IMINUSJPLUS, r + P - 1, hi_in(o.xstop), lo_in(o.ystart), lo_halo(me.xstart), hi_halo(me.ystop), n, n);
    }
    if (hasE && hasS) {
        const dlesm_region &o = subs[r - P + 1].internal;
        w.send(DLESM_IMINUSJPLUS, r - P + 1, hi_in(me.xstop), lo_in(me.ystart), lo_halo(o.xstart), hi_halo(o.ystop), n, n);
        w.recv(DLESM_IPLUSJMINUS, r - P + 1, lo_in(o.xstart), hi_in(o.ystop), hi_halo(me.xstop), lo_halo(me.ystart), n, n);
    }
    if (w.err) return fail(w.err, "ERROR: Number of separate communications exceeds maximum of %d",
                           DLESM_MAXCOMM);
    return DLESM_OK;
}
