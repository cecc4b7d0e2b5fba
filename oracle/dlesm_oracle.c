/* TEST INFRASTRUCTURE ONLY -- not part of the shipped product.  See dlesm_oracle.h.
 *
 * Plain-C restatement of the dl_esm_inf algorithms on the hot path.  Indices
 * are kept 1-based and inclusive exactly as in the Fortran so that every line
 * can be read against the reference (paths relative to
 * /root/reference/finite_difference/src).
 */
#include "dlesm_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* column-major data(1:ld, 1:ny) -> linear, field_mod.f90:350 */
#define IDX(ld, i, j) ((size_t)((j)-1) * (size_t)(ld) + (size_t)((i)-1))

/* ------------------------------------------------------------------------ */
/* grid_mod.f90:364-385 : padded leading dimension and ny                    */
void orc_grid_extents(int sub_global_nx, int sub_global_ny, int alignment, int *nx, int *ny)
{
    int a = alignment > 0 ? alignment : 1;        /* grid:350-352 default 1      */
    int padding = a - (sub_global_nx % a);        /* grid:368                    */
    *nx = sub_global_nx + padding;                /* grid:369                    */
    *ny = sub_global_ny + 1;                      /* grid:385                    */
}

/* ------------------------------------------------------------------------ */
/* field_mod.f90:563-1122                                                    */
enum { ORC_U = 0, ORC_V = 1, ORC_T = 2, ORC_F = 3, ORC_ALL = 4 };      /* field:47-52 */
enum { ORC_SW = 0, ORC_SE = 1, ORC_NW = 2, ORC_NE = 3 };                /* grid:52-57  */
enum { ORC_PERIODIC = 0 };                                              /* grid:64     */

int orc_field_bounds(int ptype, int offset, int bcx, int bcy,
                     const orc_region *s, int grid_nx, int grid_ny,
                     orc_region *in, orc_region *wh)
{
    memset(in, 0, sizeof(*in));
    memset(wh, 0, sizeof(*wh));
    if (ptype == ORC_ALL) {                       /* field_init, field:628-648   */
        in->xstart = 1; in->xstop = grid_nx;
        in->ystart = 1; in->ystop = grid_ny;
    } else if (ptype < 0 || ptype > ORC_ALL) {
        return 1;                                 /* field:591-593               */
    } else if (offset == ORC_NE) {
        /* c{u,v,t,f}_ne_init: field:755-786, 872-895, 965-1000, 1088-1122:
         * every type aborts for a periodic dimension, otherwise takes the
         * subdomain's internal region unchanged. */
        if (bcx == ORC_PERIODIC || bcy == ORC_PERIODIC) return 1;
        in->xstart = s->xstart; in->xstop = s->xstop;
        in->ystart = s->ystart; in->ystop = s->ystop;
    } else if (offset == ORC_SW) {
        in->xstart = s->xstart; in->xstop = s->xstop;
        in->ystart = s->ystart; in->ystop = s->ystop;
        switch (ptype) {
        case ORC_U:                               /* cu_sw_init field:675-751    */
            if (bcx != ORC_PERIODIC) in->xstart = s->xstart + 1;   /* field:724  */
            break;
        case ORC_V:                               /* cv_sw_init field:813-868    */
            if (bcy != ORC_PERIODIC) return 1;    /* field:844 aborts            */
            break;
        case ORC_T:                               /* ct_sw_init field:922-961    */
            break;
        case ORC_F:                               /* cf_sw_init field:1027-1084  */
            if (bcx != ORC_PERIODIC) return 1;    /* field:1048                  */
            if (bcy != ORC_PERIODIC) return 1;    /* field:1059                  */
            break;
        }
    } else {
        return 1;                                 /* field:666-668 etc.          */
    }
    in->nx = in->xstop - in->xstart + 1;          /* field:597-598               */
    in->ny = in->ystop - in->ystart + 1;
    /* field:606-619 : both branches identical, NBOUNDARY = 1 (field:227)        */
    wh->xstart = in->xstart - 1; wh->xstop = in->xstop + 1;
    wh->ystart = in->ystart - 1; wh->ystop = in->ystop + 1;
    wh->nx = wh->xstop - wh->xstart + 1;          /* field:621-622               */
    wh->ny = wh->ystop - wh->ystart + 1;
    return 0;
}

/* ------------------------------------------------------------------------ */
/* parallel_mod.f90:70-332                                                   */
void orc_decompose(int domainx, int domainy, int ndom, int ntilex_in, int ntiley_in, int hwidth,
                   orc_decomp *d, orc_subdomain *subs)
{
    int ntilex, ntiley, tmp;
    int xlen = domainx, ylen = domainy;           /* pmod:164-165                */

    d->global_nx = domainx;                       /* pmod:155-158                */
    d->global_ny = domainy;
    d->ndomains = ndom;

    if (ntilex_in <= 0) {                         /* auto_tile, pmod:167-190     */
        ntilex = (int)sqrtf((float)ndom);         /* INT(SQRT(REAL(ndom)))       */
        while (ndom % ntilex != 0) ntilex--;
        ntiley = ndom / ntilex;
        if (xlen > ylen) {
            if (ntilex < ntiley) { tmp = ntiley; ntiley = ntilex; ntilex = tmp; }
        } else {
            if (ntiley < ntilex) { tmp = ntiley; ntiley = ntilex; ntilex = tmp; }
        }
    } else {                                      /* pmod:191-194                */
        ntilex = ntilex_in;
        ntiley = ntiley_in;
    }
    d->nx = ntilex;                               /* pmod:199-200                */
    d->ny = ntiley;

    int internal_width = xlen / ntilex;           /* pmod:204-205                */
    int internal_height = ylen / ntiley;
    int nwidth = ntiley * internal_height;        /* pmod:211-216                */
    int junder = nwidth < ylen ? ylen - nwidth : 0;
    nwidth = ntilex * internal_width;             /* pmod:218-223                */
    int iunder = nwidth < xlen ? xlen - nwidth : 0;

    int ith = 0, jval = 1;                        /* pmod:232-234 (ith 0-based)  */
    d->max_width = 0;
    d->max_height = 0;
    for (int jj = 1; jj <= ntiley; jj++) {        /* pmod:244                    */
        int height;
        if (junder > 0) { height = internal_height + 1; junder--; }   /* pmod:251-256 */
        else height = internal_height;
        int ival = 1, iunder_row = iunder;        /* pmod:259-260                */
        orc_subdomain *sd = NULL;
        for (int ji = 1; ji <= ntilex; ji++) {    /* pmod:262                    */
            int width;
            if (iunder_row > 0) { width = internal_width + 1; iunder_row--; } /* pmod:265-270 */
            else width = internal_width;
            sd = &subs[ith];
            sd->internal.xstart = hwidth + 1;                       /* pmod:273  */
            sd->internal.xstop = sd->internal.xstart + width - 1;
            sd->internal.nx = width;
            sd->global.xstart = ival;                               /* pmod:278  */
            sd->global.xstop = sd->global.xstart + width - 1;
            sd->global.nx = 2 * hwidth + width;                     /* pmod:281  */
            sd->internal.ystart = hwidth + 1;                       /* pmod:283  */
            sd->internal.ystop = sd->internal.ystart + height - 1;
            sd->internal.ny = height;
            sd->global.ystart = jval;                               /* pmod:287  */
            sd->global.ystop = sd->global.ystart + height - 1;
            sd->global.ny = 2 * hwidth + sd->internal.ny;           /* pmod:290  */
            if (sd->global.nx > d->max_width) d->max_width = sd->global.nx;   /* pmod:310 */
            if (sd->global.ny > d->max_height) d->max_height = sd->global.ny;
            ival = sd->global.xstop + 1;                            /* pmod:313  */
            ith++;
        }
        jval = sd->global.ystop + 1;                                /* pmod:316  */
    }
}

/* ------------------------------------------------------------------------ */
/* parallel_comms_mod.f90:1365-1398                                          */
int orc_iprocmap(const orc_decomp *d, const orc_subdomain *subs, int nranks, int ia, int ja)
{
    (void)d;
    for (int iproc = 1; iproc <= nranks; iproc++) {
        const orc_region *g = &subs[iproc - 1].global;
        if (g->xstart <= ia && ia <= g->xstop && g->ystart <= ja && ja <= g->ystop) return iproc;
    }
    return 0;
}

/* direction codes, parallel_comms_mod.f90:101-123 */
enum { Iplus = 1, Iminus = 2, Jplus = 3, Jminus = 4,
       IplusJplus = 5, IminusJminus = 6, IplusJminus = 7, IminusJplus = 8 };
static const int west_[9]  = {0, 1, 0, 0, 0, 1, 0, 1, 0};
static const int east_[9]  = {0, 0, 1, 0, 0, 0, 1, 0, 1};
static const int south_[9] = {0, 0, 0, 1, 0, 1, 0, 0, 1};
static const int north_[9] = {0, 0, 0, 0, 1, 0, 1, 1, 0};
static const int opp_dirn[9] = {0, Iminus, Iplus, Jminus, Jplus,          /* pcomms:237-244 */
                                IminusJminus, IplusJplus, IminusJplus, IplusJminus};

/* pcomms:1174-1267 */
static int addsend(orc_comms *c, int dir, int proc, int isrc, int jsrc, int ides, int jdes,
                   int nx, int ny)
{
    if (proc < 0) return 0;
    if (c->nsend + 1 > ORC_MAXCOMM) return -12;
    int k = c->nsend++;
    c->dirsend[k] = dir; c->destination[k] = proc;
    c->isrcsend[k] = isrc; c->jsrcsend[k] = jsrc;
    c->idessend[k] = ides; c->jdessend[k] = jdes;
    c->nxsend[k] = nx; c->nysend[k] = ny;
    return 0;
}

/* pcomms:1269-1363 */
static int addrecv(orc_comms *c, int dir, int proc, int isrc, int jsrc, int ides, int jdes,
                   int nx, int ny)
{
    if (proc < 0) return 0;
    if (c->nrecv + 1 > ORC_MAXCOMM) return -12;
    int k = c->nrecv++;
    c->dirrecv[k] = dir; c->source[k] = proc;
    c->isrcrecv[k] = isrc; c->jsrcrecv[k] = jsrc;
    c->idesrecv[k] = ides; c->jdesrecv[k] = jdes;
    c->nxrecv[k] = nx; c->nyrecv[k] = ny;
    return 0;
}

/* parallel_comms_mod.f90:178-1172 with halo_depthx = halo_depthy = 1 (the only
 * value the reference supports, pcomms:48,220-223), so every "do ihalo" loop
 * has the single iteration ihalo = 1. */
int orc_map_comms(const orc_decomp *d, const orc_subdomain *subs, int nranks, int irank,
                  orc_comms *c)
{
#define PM(ia, ja) orc_iprocmap(d, subs, nranks, (ia), (ja))
    const orc_subdomain *me = &subs[irank - 1];
    const int ihalo = 1;
    int ierr;
    memset(c, 0, sizeof(*c));
    for (int k = 0; k < ORC_MAXCOMM; k++) {       /* pcomms:246-261 */
        c->dirsend[k] = c->destination[k] = c->isrcsend[k] = c->jsrcsend[k] = -999;
        c->idessend[k] = c->jdessend[k] = c->nxsend[k] = c->nysend[k] = -999;
        c->dirrecv[k] = c->source[k] = c->isrcrecv[k] = c->jsrcrecv[k] = -999;
        c->idesrecv[k] = c->jdesrecv[k] = c->nxrecv[k] = c->nyrecv[k] = -999;
    }
    const int jelb = me->global.ystart, jeub = me->global.ystop;   /* pcomms:278-283 */
    const int ielb = me->global.xstart, ieub = me->global.xstop;

    int isrcs, jsrcs, idess, jdess, isrcr, jsrcr, idesr, jdesr, nxs, nys, nxr, nyr;
    int naddmaxs, naddmaxr, nadd;

    /* --- send in minus I (Iplus), receive what was sent in plus I: pcomms:296-471 */
    int j1 = jelb;
    while (j1 <= jeub) {
        int iproc = PM(ielb - 1, j1);
        if (iproc > 0) {
            const orc_subdomain *o = &subs[iproc - 1];
            int j2 = jeub < o->global.ystop ? jeub : o->global.ystop;       /* :307 */
            isrcs = me->internal.xstart;                                     /* :328 */
            isrcr = o->internal.xstop - ihalo + 1;                           /* :333 */
            idesr = ihalo; nxr = ihalo; nxs = ihalo;                         /* :334-336 */
            idess = o->internal.xstop + 1;                                   /* :340 */
            jsrcs = j1 - me->global.ystart + me->internal.ystart;            /* :342 */
            jdess = j1 - o->global.ystart + o->internal.ystart;              /* :343 */
            jdesr = jsrcs; jsrcr = jdess;                                    /* :345-346 */
            nyr = j2 - j1 + 1; nys = nyr;                                    /* :347-348 */
            naddmaxr = naddmaxs = 0;                                         /* :352-372 */
            if (j1 - ihalo >= jelb && PM(ielb - ihalo, j1) > 0) naddmaxs = ihalo;
            if (j1 == jelb && PM(ielb - ihalo, j1 - ihalo) == iproc) naddmaxr = ihalo;
            nadd = ihalo < naddmaxs ? ihalo : naddmaxs;                      /* :376-390 */
            jdess -= nadd; jsrcs -= nadd; nys += nadd;
            nadd = ihalo < naddmaxr ? ihalo : naddmaxr;
            jdesr -= nadd; jsrcr -= nadd; nyr += nadd;
            naddmaxr = naddmaxs = 0;                                         /* :401-420 */
            if (j2 + ihalo <= jeub && PM(ielb - ihalo, j2) > 0) naddmaxs = ihalo;
            if (j2 == jeub && PM(ielb - ihalo, j2 + ihalo) == iproc) naddmaxr = ihalo;
            nys += ihalo < naddmaxs ? ihalo : naddmaxs;                      /* :424-441 */
            nyr += ihalo < naddmaxr ? ihalo : naddmaxr;
            if ((ierr = addsend(c, Iplus, iproc - 1, isrcs, jsrcs, idess, jdess, nxs, nys))) return ierr;
            if ((ierr = addrecv(c, Iminus, iproc - 1, isrcr, jsrcr, idesr, jdesr, nxr, nyr))) return ierr;
            j1 = j2 + 1;                                                     /* :464 */
        } else {
            j1++;                                                            /* :469 */
        }
    }

    /* --- send in plus I (Iminus), receive what was sent in minus I: pcomms:483-636 */
    j1 = jelb;
    while (j1 <= jeub) {
        int iproc = PM(ieub + 1, j1);
        if (iproc > 0) {
            const orc_subdomain *o = &subs[iproc - 1];
            int j2 = jeub < o->global.ystop ? jeub : o->global.ystop;       /* :494 */
            isrcr = o->internal.xstart + ihalo;    /* :521 (as written in the reference) */
            isrcs = me->internal.xstop - ihalo + 1;                          /* :523 */
            idess = ihalo; nxr = ihalo; nxs = ihalo;                         /* :524-526 */
            idesr = me->internal.xstop + ihalo;                              /* :528 */
            jdess = j1 - o->global.ystart + o->internal.ystart;              /* :531 */
            jsrcs = j1 - me->global.ystart + me->internal.ystart;            /* :534 */
            jdesr = jsrcs; jsrcr = jdess;                                    /* :536-537 */
            nyr = j2 - j1 + 1; nys = nyr;                                    /* :538-539 */
            naddmaxr = naddmaxs = 0;                                         /* :542-560 */
            if (j1 - ihalo >= jelb && PM(ieub + ihalo, j1) > 0) naddmaxs = ihalo;
            if (j1 == jelb && PM(ieub + ihalo, j1 - ihalo) == iproc) naddmaxr = ihalo;
            nadd = ihalo < naddmaxs ? ihalo : naddmaxs;                      /* :563-572 */
            jdess -= nadd; jsrcs -= nadd; nys += nadd;
            nadd = ihalo < naddmaxr ? ihalo : naddmaxr;
            jdesr -= nadd; jsrcr -= nadd; nyr += nadd;
            naddmaxr = naddmaxs = 0;                                         /* :575-594 */
            if (j2 + ihalo <= jeub && PM(ieub + ihalo, j2) > 0) naddmaxs = ihalo;
            if (j2 == jeub && PM(ieub + ihalo, j2 + ihalo) == iproc) naddmaxr = ihalo;
            nys += ihalo < naddmaxs ? ihalo : naddmaxs;                      /* :598-615 */
            nyr += ihalo < naddmaxr ? ihalo : naddmaxr;
            if ((ierr = addsend(c, Iminus, iproc - 1, isrcs, jsrcs, idess, jdess, nxs, nys))) return ierr;
            if ((ierr = addrecv(c, Iplus, iproc - 1, isrcr, jsrcr, idesr, jdesr, nxr, nyr))) return ierr;
            j1 = j2 + 1;
        } else {
            j1++;
        }
    }

    /* --- send in minus J (Jplus), receive what was sent in plus J: pcomms:648-818 */
    const int imin = ielb, imax = ieub;
    int i1 = imin;
    while (i1 <= imax) {
        int iproc = PM(i1, jelb - 1);
        if (iproc > 0) {
            const orc_subdomain *o = &subs[iproc - 1];
            int i2 = imax < o->global.xstop ? imax : o->global.xstop;       /* :662-666 */
            isrcs = i1 - me->global.xstart + me->internal.xstart;            /* :685 */
            idess = i1 - o->global.xstart + o->internal.xstart;              /* :688 */
            idesr = isrcs; isrcr = idess;                                    /* :690-691 */
            nxr = i2 - i1 + 1; nxs = nxr;                                    /* :692-693 */
            jsrcs = me->internal.ystart;                                     /* :695 */
            jsrcr = o->internal.ystop - ihalo + 1;                           /* :699 */
            jdesr = ihalo; nyr = ihalo; nys = ihalo;                         /* :700-702 */
            jdess = o->internal.ystop + 1;                                   /* :705 */
            naddmaxr = naddmaxs = 0;                                         /* :708-725 */
            if (i1 - ihalo >= imin && PM(i1 - ihalo, jelb - ihalo) > 0) naddmaxs = ihalo;
            if (i1 == imin && PM(i1 - ihalo, jelb - ihalo) == iproc) naddmaxr = ihalo;
            nadd = ihalo < naddmaxs ? ihalo : naddmaxs;                      /* :729-751 */
            idess -= nadd; isrcs -= nadd; nxs += nadd;
            nadd = ihalo < naddmaxr ? ihalo : naddmaxr;
            idesr -= nadd; isrcr -= nadd; nxr += nadd;
            naddmaxr = naddmaxs = 0;                                         /* :755-774 */
            if (i2 + ihalo <= imax && PM(i2, jelb - ihalo) > 0) naddmaxs = ihalo;
            if (i2 == imax && PM(i2 + ihalo, jelb - ihalo) == iproc) naddmaxr = ihalo;
            nxs += ihalo < naddmaxs ? ihalo : naddmaxs;                      /* :778-795 */
            nxr += ihalo < naddmaxr ? ihalo : naddmaxr;
            if ((ierr = addsend(c, Jplus, iproc - 1, isrcs, jsrcs, idess, jdess, nxs, nys))) return ierr;
            if ((ierr = addrecv(c, Jminus, iproc - 1, isrcr, jsrcr, idesr, jdesr, nxr, nyr))) return ierr;
            i1 = i2 + 1;
        } else {
            i1++;
        }
    }

    /* --- send in plus J (Jminus), receive what was sent in minus J: pcomms:829-986 */
    i1 = imin;
    while (i1 <= imax) {
        int iproc = PM(i1, jeub + 1);
        if (iproc > 0) {
            const orc_subdomain *o = &subs[iproc - 1];
            int i2 = imax < o->global.xstop ? imax : o->global.xstop;       /* :845 */
            isrcs = i1 - me->global.xstart + me->internal.xstart;            /* :854 */
            idess = i1 - o->global.xstart + o->internal.xstart;              /* :855 */
            idesr = isrcs; isrcr = idess;                                    /* :857-858 */
            nxr = i2 - i1 + 1; nxs = nxr;                                    /* :859-860 */
            jsrcr = o->internal.ystart;                                      /* :863 */
            jsrcs = me->internal.ystop - ihalo + 1;                          /* :867 */
            jdess = ihalo; nyr = ihalo; nys = ihalo;                         /* :868-870 */
            jdesr = me->internal.ystop + 1;                                  /* :873 */
            naddmaxr = naddmaxs = 0;                                         /* :877-894 */
            if (i1 - ihalo >= imin && PM(i1, jeub + ihalo) > 0) naddmaxs = ihalo;
            if (i1 == imin && PM(i1 - ihalo, jeub + ihalo) == iproc) naddmaxr = ihalo;
            nadd = ihalo < naddmaxs ? ihalo : naddmaxs;                      /* :898-920 */
            idess -= nadd; isrcs -= nadd; nxs += nadd;
            nadd = ihalo < naddmaxr ? ihalo : naddmaxr;
            idesr -= nadd; isrcr -= nadd; nxr += nadd;
            naddmaxr = naddmaxs = 0;                                         /* :923-941 */
            if (i2 + ihalo <= imax && PM(i2, jeub + ihalo) > 0) naddmaxs = ihalo;
            if (i2 == imax && PM(i2 + ihalo, jeub + ihalo) == iproc) naddmaxr = ihalo;
            nxs += ihalo < naddmaxs ? ihalo : naddmaxs;                      /* :945-964 */
            nxr += ihalo < naddmaxr ? ihalo : naddmaxr;
            if ((ierr = addsend(c, Jminus, iproc - 1, isrcs, jsrcs, idess, jdess, nxs, nys))) return ierr;
            if ((ierr = addrecv(c, Jplus, iproc - 1, isrcr, jsrcr, idesr, jdesr, nxr, nyr))) return ierr;
            i1 = i2 + 1;
        } else {
            i1++;
        }
    }

    /* --- diagonal messages: pcomms:998-1170 */
    for (int idirn = 5; idirn <= 8; idirn++) {
        const int w = west_[idirn], e = east_[idirn], s = south_[idirn], n = north_[idirn];
        int addcorner = 0;
        int ioutside = w * (me->global.xstart - 1) + e * (me->global.xstop + 1);  /* :1005 */
        int iinside = w * me->global.xstart + e * me->global.xstop;               /* :1007 */
        int iprocx = PM(ioutside, s * jelb + n * jeub);                            /* :1019 */
        int iprocy = PM(iinside, s * (jelb - 1) + n * (jeub + 1));                 /* :1022 */
        int iproc = PM(ioutside - w * (ihalo - 1) + e * (ihalo - 1),               /* :1033 */
                       s * (jelb - ihalo) + n * (jeub + ihalo));
        if (iproc > 0 && iprocx > 0 && iprocy > 0 && iproc != iprocx && iproc != iprocy) { /* :1039 */
            const orc_subdomain *o = &subs[iproc - 1];
            int ielb_iproc = o->global.xstart, ieub_iproc = o->global.xstop;
            int jelb_iproc = o->global.ystart, jeub_iproc = o->global.ystop;
            addcorner = 1;                                                         /* :1055-1061 */
            int ldiff0 = ielb_iproc - ieub, ldiff1 = ielb - ieub_iproc;            /* :1068 */
            nxs = ihalo - e * (ldiff0 - 1) - w * (ldiff1 - 1);
            ldiff0 = jelb_iproc - jeub; ldiff1 = jelb - jeub_iproc;               /* :1073 */
            nys = ihalo - n * (ldiff0 - 1) - s * (ldiff1 - 1);
            isrcs = e * (me->internal.xstop - ihalo + 1) + w * (me->internal.xstart + ihalo - 1);
            jsrcs = n * (me->internal.ystop - ihalo + 1) + s * (me->internal.ystart + ihalo - 1);
            idess = w * (o->internal.xstop + ihalo) + e * (o->internal.xstart - ihalo);   /* :1087 */
            jdess = s * (o->internal.ystop + ihalo) + n * (o->internal.ystart - ihalo);   /* :1091 */
            isrcr = w * (o->internal.xstop - ihalo + 1) + e * (o->internal.xstart + ihalo - 1);
            jsrcr = s * (o->internal.ystop - ihalo + 1) + n * (o->internal.ystart + ihalo - 1);
            idesr = e * (me->internal.xstop + ihalo) + w * (me->internal.xstart - ihalo); /* :1107 */
            jdesr = n * (me->internal.ystop + ihalo) + s * (me->internal.ystart - ihalo); /* :1110 */
        } else {                                                                    /* :1120-1132 */
            isrcs = jsrcs = idess = jdess = isrcr = jsrcr = idesr = jdesr = 0;
            nxs = nys = 0;
        }
        nxr = nxs; nyr = nys;                                                      /* :1139-1140 */
        if (addcorner) {                                                            /* :1143-1168 */
            if ((ierr = addsend(c, idirn, iproc - 1, isrcs, jsrcs, idess, jdess, nxs, nys))) return ierr;
            if ((ierr = addrecv(c, opp_dirn[idirn], iproc - 1, isrcr, jsrcr, idesr, jdesr, nxr, nyr)))
                return ierr;
        }
    }
    return 0;
#undef PM
}

/* ------------------------------------------------------------------------ */
/* exchange_generic for every rank at once: pack pcomms:1664-1691, unpack
 * pcomms:1773-1798; all four comm directions enabled as field:1247-1248 does,
 * hence all diagonals enabled too (pcomms:1568-1571). */
static int orc_exchange_enabled(int nranks, double **fields, const int *ld, const orc_comms *comms,
                                const int *enabled);

int orc_exchange_all(int nranks, double **fields, const int *ld, const orc_comms *comms)
{
    const int enabled[9] = {0, 1, 1, 1, 1, 1, 1, 1, 1};
    return orc_exchange_enabled(nranks, fields, ld, comms, enabled);
}

/* exchange_generic with an explicit choice of comm1..comm4 (pcomms:1557-1571): a direction is
 * enabled when one of the four arguments names it; a diagonal when both its edges are.  With
 * no_diagonals != 0 the diagonals stay off whatever the edges say -- no reference call does
 * that; it states what the distributed 5-point step exchanges (edges only). */
int orc_exchange_dirs(int nranks, double **fields, const int *ld, const orc_comms *comms,
                      int comm1, int comm2, int comm3, int comm4, int no_diagonals)
{
    int enabled[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    const int c[4] = {comm1, comm2, comm3, comm4};
    for (int k = 0; k < 4; k++)
        if (c[k] >= 0 && c[k] <= 8) enabled[c[k]] = c[k] > 0;                    /* :1561-1564 */
    enabled[5] = enabled[1] && enabled[3];                                       /* IplusJplus   :1568 */
    enabled[6] = enabled[2] && enabled[4];                                       /* IminusJminus :1569 */
    enabled[7] = enabled[1] && enabled[4];                                       /* IplusJminus  :1570 */
    enabled[8] = enabled[2] && enabled[3];                                       /* IminusJplus  :1571 */
    if (no_diagonals) enabled[5] = enabled[6] = enabled[7] = enabled[8] = 0;
    return orc_exchange_enabled(nranks, fields, ld, comms, enabled);
}

static int orc_exchange_enabled(int nranks, double **fields, const int *ld, const orc_comms *comms,
                                const int *enabled)
{
    int unmatched = 0;
    /* stage every send buffer first (sendBuff(ic,isend) = b2(i,j), j outer / i inner) */
    double **buf = (double **)calloc((size_t)nranks * ORC_MAXCOMM, sizeof(double *));
    for (int r = 0; r < nranks; r++) {
        const orc_comms *c = &comms[r];
        for (int s = 0; s < c->nsend; s++) {
            if (!enabled[c->dirsend[s]]) continue;                                /* :1639 */
            if (c->destination[s] < 0 || c->nxsend[s] <= 0) continue;            /* :1639-1640 */
            size_t n = (size_t)c->nxsend[s] * (size_t)c->nysend[s], ic = 0;
            double *b = (double *)malloc(n * sizeof(double));
            int istart = c->isrcsend[s], iend = istart + c->nxsend[s] - 1;       /* :1668-1671 */
            int jstart = c->jsrcsend[s], jend = jstart + c->nysend[s] - 1;
            for (int j = jstart; j <= jend; j++)
                for (int i = istart; i <= iend; i++) b[ic++] = fields[r][IDX(ld[r], i, j)];
            buf[(size_t)r * ORC_MAXCOMM + s] = b;
        }
    }
    /* deliver: receiver's (source, dirrecv) selects the sender's (destination, dirsend):
     * tag = tag_orig + dir on both sides (pcomms:1606,1647) */
    for (int r = 0; r < nranks; r++) {
        const orc_comms *c = &comms[r];
        for (int q = 0; q < c->nrecv; q++) {
            if (!enabled[c->dirrecv[q]]) continue;                                /* :1603 */
            if (c->source[q] < 0 || c->nxrecv[q] <= 0) continue;                 /* :1603-1604 */
            int src = c->source[q];
            const orc_comms *cs = &comms[src];
            int found = -1;
            for (int s = 0; s < cs->nsend; s++)
                if (cs->destination[s] == r && cs->dirsend[s] == c->dirrecv[q]) { found = s; break; }
            if (found < 0 || cs->nxsend[found] * cs->nysend[found] != c->nxrecv[q] * c->nyrecv[q]) {
                unmatched++;
                continue;
            }
            const double *b = buf[(size_t)src * ORC_MAXCOMM + found];
            size_t ic = 0;
            int jstart = c->jdesrecv[q], jend = jstart + c->nyrecv[q] - 1;       /* :1777-1781 */
            int istart = c->idesrecv[q], iend = istart + c->nxrecv[q] - 1;
            for (int j = jstart; j <= jend; j++)
                for (int i = istart; i <= iend; i++) fields[r][IDX(ld[r], i, j)] = b[ic++];
        }
    }
    for (size_t k = 0; k < (size_t)nranks * ORC_MAXCOMM; k++) free(buf[k]);
    free(buf);
    return unmatched;
}

/* ------------------------------------------------------------------------ */
/* field_mod.f90:1298-1302 ; accumulated in extended precision so that the
 * oracle is the reference value a tree-reduction on the device is compared to
 * (the Fortran SUM order is compiler-defined). */
double orc_checksum(const double *f, int ld, int xstart, int xstop, int ystart, int ystop)
{
    long double acc = 0.0L;
    for (int j = ystart; j <= ystop; j++)
        for (int i = xstart; i <= xstop; i++) acc += (long double)fabs(f[IDX(ld, i, j)]);
    return (double)acc;
}

/* field_mod.f90:378-389 */
void orc_scatter(const double *global, int gnx, const orc_subdomain *sub, double *local, int ld)
{
    int dx = sub->global.xstart - sub->internal.xstart;                          /* :379 */
    int dy = sub->global.ystart - sub->internal.ystart;                          /* :380 */
    for (int jj = sub->internal.ystart; jj <= sub->internal.ystop; jj++)
        for (int ji = sub->internal.xstart; ji <= sub->internal.xstop; ji++)
            local[IDX(ld, ji, jj)] = global[IDX(gnx, ji + dx, jj + dy)];         /* :386 */
}

/* field_mod.f90:1313-1390 ; T-point field (internal == subdomain internal) */
void orc_gather_all(int nranks, double **fields, const int *ld, const orc_decomp *d,
                    const orc_subdomain *subs, double *global)
{
    for (int r = 0; r < nranks; r++) {
        const orc_subdomain *s = &subs[r];
        /* send_buffer in j-outer/i-inner order (:1362-1368), unpacked in the same
         * order into the rank's global box (:1376-1386) */
        int gj = s->global.ystart;
        for (int jj = s->internal.ystart; jj <= s->internal.ystop; jj++, gj++) {
            int gi = s->global.xstart;
            for (int ji = s->internal.xstart; ji <= s->internal.xstop; ji++, gi++)
                global[IDX(d->global_nx, gi, gj)] = fields[r][IDX(ld[r], ji, jj)];
        }
    }
}

/* ------------------------------------------------------------------------ */
/* splitmix64 (public-domain constant set) -> 53-bit uniform in [0,1) */
double orc_hash_u01(uint64_t seed, int64_t gi, int64_t gj)
{
    uint64_t x = seed ^ ((uint64_t)gi + ((uint64_t)gj << 32));
    x += 0x9E3779B97F4A7C15ULL;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ULL;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBULL;
    x = x ^ (x >> 31);
    return (double)(x >> 11) * 0x1.0p-53;
}

/* ------------------------------------------------------------------------ */
/* GOcean kernel form: one call per point from the PSy-layer double loop
 * (infrastructure_mod.f90:32-41, field_mod.f90:343-349). */
static inline void jacobi5_code(int ji, int jj, double *out, const double *in, int ld)
{
    out[IDX(ld, ji, jj)] = 0.25 * ((in[IDX(ld, ji - 1, jj)] + in[IDX(ld, ji + 1, jj)]) +
                                   (in[IDX(ld, ji, jj - 1)] + in[IDX(ld, ji, jj + 1)]));
}

void orc_jacobi5(const double *in, double *out, int ld,
                 int xstart, int xstop, int ystart, int ystop)
{
    for (int jj = ystart; jj <= ystop; jj++)
        for (int ji = xstart; ji <= xstop; ji++) jacobi5_code(ji, jj, out, in, ld);
}

/* The same PSy loop nest for a kernel that requests the T mask (GO_GRID_MASK_T,
 * argument_mod.f90:75-112): specification frozen in DESIGN.md section 5.7 -- dry points carry
 * their value over, a dry neighbour is mirrored (no-flux coast).  tmask has the field layout. */
static inline void jacobi5_masked_code(int ji, int jj, double *out, const double *in, const int *tmask, int ld)
{
    const double c = in[IDX(ld, ji, jj)];
    if (tmask[IDX(ld, ji, jj)] <= 0) {
        out[IDX(ld, ji, jj)] = c;
        return;
    }
    const double w = tmask[IDX(ld, ji - 1, jj)] > 0 ? in[IDX(ld, ji - 1, jj)] : c;
    const double e = tmask[IDX(ld, ji + 1, jj)] > 0 ? in[IDX(ld, ji + 1, jj)] : c;
    const double s = tmask[IDX(ld, ji, jj - 1)] > 0 ? in[IDX(ld, ji, jj - 1)] : c;
    const double n = tmask[IDX(ld, ji, jj + 1)] > 0 ? in[IDX(ld, ji, jj + 1)] : c;
    out[IDX(ld, ji, jj)] = 0.25 * ((w + e) + (s + n));
}

void orc_jacobi5_masked(const double *in, double *out, const int *tmask, int ld,
                        int xstart, int xstop, int ystart, int ystop)
{
    for (int jj = ystart; jj <= ystop; jj++)
        for (int ji = xstart; ji <= xstop; ji++) jacobi5_masked_code(ji, jj, out, in, tmask, ld);
}

/* grid_init's T mask, grid_mod.f90:394-455: the grid's own (nx, ny) copy of the user's mask --
 * copy-in of the subdomain plus its one-cell ring (:407-412), then the rows beyond the ring take
 * the ring row (:416-422) and the columns beyond it the ring column (:424-430).  user == NULL:
 * the all-wet mask on the same region (:447-453); cells the reference leaves unset are 0 here.
 * `user` has leading dimension user_ld (the subdomain's whole width, as the callers allocate it). */
void orc_tmask_fill(const int *user, int user_ld, int nx, int ny, int xstart, int xstop, int ystart,
                    int ystop, int *tmask)
{
    for (long k = 0; k < (long)nx * ny; k++) tmask[k] = 0;
    for (int jj = ystart - 1; jj <= ystop + 1; jj++)
        for (int ji = xstart - 1; ji <= xstop + 1; ji++)
            tmask[IDX(nx, ji, jj)] = user ? user[IDX(user_ld, ji, jj)] : 1;
    if (!user) return;
    for (int jj = ystop + 2; jj <= ny; jj++)                                  /* "North" :416-418 */
        for (int ji = 1; ji <= nx; ji++) tmask[IDX(nx, ji, jj)] = tmask[IDX(nx, ji, ystop + 1)];
    for (int jj = 1; jj <= ystart - 2; jj++)                                  /* "South" :420-422 */
        for (int ji = 1; ji <= nx; ji++) tmask[IDX(nx, ji, jj)] = tmask[IDX(nx, ji, ystart - 1)];
    for (int ji = 1; ji <= xstart - 2; ji++)                                  /* :424-426 */
        for (int jj = 1; jj <= ny; jj++) tmask[IDX(nx, ji, jj)] = tmask[IDX(nx, xstart - 1, jj)];
    for (int ji = xstop + 2; ji <= nx; ji++)                                  /* :428-430 */
        for (int jj = 1; jj <= ny; jj++) tmask[IDX(nx, ji, jj)] = tmask[IDX(nx, xstop + 1, jj)];
}

/* A general 3x3 weighted stencil in GOcean kernel form; coef[(dj+1)*3 + (di+1)].  Evaluation order
 * frozen in DESIGN.md section 5.9: rows south to north, each row west to east. */
static inline void stencil9_code(int ji, int jj, double *out, const double *in, const double *c, int ld)
{
    const double S = (c[0] * in[IDX(ld, ji - 1, jj - 1)] + c[1] * in[IDX(ld, ji, jj - 1)]) + c[2] * in[IDX(ld, ji + 1, jj - 1)];
    const double M = (c[3] * in[IDX(ld, ji - 1, jj)] + c[4] * in[IDX(ld, ji, jj)]) + c[5] * in[IDX(ld, ji + 1, jj)];
    const double N = (c[6] * in[IDX(ld, ji - 1, jj + 1)] + c[7] * in[IDX(ld, ji, jj + 1)]) + c[8] * in[IDX(ld, ji + 1, jj + 1)];
    out[IDX(ld, ji, jj)] = (S + M) + N;
}

void orc_stencil9(const double *in, double *out, const double *coef, int ld,
                  int xstart, int xstop, int ystart, int ystop)
{
    for (int jj = ystart; jj <= ystop; jj++)
        for (int ji = xstart; ji <= xstop; ji++) stencil9_code(ji, jj, out, in, coef, ld);
}

/* Free-surface (continuity) update of a NEMOLite2D-class model in GOcean kernel form: fields on T, U and V
 * points plus the grid property area_t (GO_GRID_AREA_T, argument_mod.f90:75-112).  Specification frozen in
 * DESIGN.md section 5.10; the Fortran expression `(rtmp2 - rtmp1 + rtmp4 - rtmp3) * rdt / e12t` evaluates
 * left to right.  PARITY UNPINNED by the reference (it holds no such kernel). */
static inline void continuity_code(int ji, int jj, double *ssha, const double *sshn_t, const double *sshn_u,
                                   const double *sshn_v, const double *hu, const double *hv, const double *un,
                                   const double *vn, double rdt, const double *area_t, int ld)
{
    const double rtmp1 = (sshn_u[IDX(ld, ji, jj)] + hu[IDX(ld, ji, jj)]) * un[IDX(ld, ji, jj)];
    const double rtmp2 = (sshn_u[IDX(ld, ji - 1, jj)] + hu[IDX(ld, ji - 1, jj)]) * un[IDX(ld, ji - 1, jj)];
    const double rtmp3 = (sshn_v[IDX(ld, ji, jj)] + hv[IDX(ld, ji, jj)]) * vn[IDX(ld, ji, jj)];
    const double rtmp4 = (sshn_v[IDX(ld, ji, jj - 1)] + hv[IDX(ld, ji, jj - 1)]) * vn[IDX(ld, ji, jj - 1)];
    ssha[IDX(ld, ji, jj)] = sshn_t[IDX(ld, ji, jj)] + (rtmp2 - rtmp1 + rtmp4 - rtmp3) * rdt / area_t[IDX(ld, ji, jj)];
}

void orc_continuity(double rdt, int ld, int xstart, int xstop, int ystart, int ystop, const double *sshn_t,
                    const double *sshn_u, const double *sshn_v, const double *hu, const double *hv, const double *un,
                    const double *vn, const double *area_t, double *ssha)
{
    for (int jj = ystart; jj <= ystop; jj++)
        for (int ji = xstart; ji <= xstop; ji++)
            continuity_code(ji, jj, ssha, sshn_t, sshn_u, sshn_v, hu, hv, un, vn, rdt, area_t, ld);
}

void orc_jacobi5_omp(const double *in, double *out, int ld,
                     int xstart, int xstop, int ystart, int ystop, int nthreads)
{
    (void)nthreads;
#pragma omp parallel for schedule(static) num_threads(nthreads)
    for (int jj = ystart; jj <= ystop; jj++)
        for (int ji = xstart; ji <= xstop; ji++) jacobi5_code(ji, jj, out, in, ld);
}

/* row-parallel copy with the same static schedule as orc_jacobi5_omp: used by the
 * cpu_baseline leg to first-touch its input on the threads that will read it */
void orc_copy_rows_omp(double *dst, const double *src, int ld, int ny, int nthreads)
{
    (void)nthreads;
#pragma omp parallel for schedule(static) num_threads(nthreads)
    for (int jj = 0; jj < ny; jj++) memcpy(dst + (size_t)jj * ld, src + (size_t)jj * ld, (size_t)ld * sizeof(double));
}

int orc_max_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* ------------------------------------------------------------------------ */
/* Shallow-water kernels, NE offset: u(i,j) is the east face of T(i,j), v(i,j)
 * the north face, z(i,j) the NE corner.  Formulas frozen in DESIGN.md section 6. */
#define A(f, i, j) (f)[IDX(ld, (i), (j))]

static inline void compute_cu_code(int i, int j, double *cu, const double *p, const double *u, int ld)
{
    A(cu, i, j) = 0.5 * (A(p, i + 1, j) + A(p, i, j)) * A(u, i, j);
}
static inline void compute_cv_code(int i, int j, double *cv, const double *p, const double *v, int ld)
{
    A(cv, i, j) = 0.5 * (A(p, i, j + 1) + A(p, i, j)) * A(v, i, j);
}
static inline void compute_z_code(int i, int j, double *z, const double *p, const double *u,
                                  const double *v, double fsdx, double fsdy, int ld)
{
    A(z, i, j) = (fsdx * (A(v, i + 1, j) - A(v, i, j)) - fsdy * (A(u, i, j + 1) - A(u, i, j))) /
                 (A(p, i, j) + A(p, i + 1, j) + A(p, i + 1, j + 1) + A(p, i, j + 1));
}
static inline void compute_h_code(int i, int j, double *h, const double *p, const double *u,
                                  const double *v, int ld)
{
    A(h, i, j) = A(p, i, j) + 0.25 * (A(u, i, j) * A(u, i, j) + A(u, i - 1, j) * A(u, i - 1, j) +
                                      A(v, i, j) * A(v, i, j) + A(v, i, j - 1) * A(v, i, j - 1));
}
static inline void compute_unew_code(int i, int j, double *unew, const double *uold, const double *z,
                                     const double *cv, const double *h, double tdts8, double tdtsdx,
                                     int ld)
{
    A(unew, i, j) = A(uold, i, j) +
                    tdts8 * (A(z, i, j) + A(z, i, j - 1)) *
                        (A(cv, i + 1, j) + A(cv, i, j) + A(cv, i, j - 1) + A(cv, i + 1, j - 1)) -
                    tdtsdx * (A(h, i + 1, j) - A(h, i, j));
}
static inline void compute_vnew_code(int i, int j, double *vnew, const double *vold, const double *z,
                                     const double *cu, const double *h, double tdts8, double tdtsdy,
                                     int ld)
{
    A(vnew, i, j) = A(vold, i, j) -
                    tdts8 * (A(z, i, j) + A(z, i - 1, j)) *
                        (A(cu, i, j + 1) + A(cu, i - 1, j + 1) + A(cu, i - 1, j) + A(cu, i, j)) -
                    tdtsdy * (A(h, i, j + 1) - A(h, i, j));
}
static inline void compute_pnew_code(int i, int j, double *pnew, const double *pold, const double *cu,
                                     const double *cv, double tdtsdx, double tdtsdy, int ld)
{
    A(pnew, i, j) = A(pold, i, j) - tdtsdx * (A(cu, i, j) - A(cu, i - 1, j)) -
                    tdtsdy * (A(cv, i, j) - A(cv, i, j - 1));
}

void orc_sw_step(const orc_sw_params *q, int ld, int xs, int xe, int ys, int ye,
                 const double *u, const double *v, const double *p,
                 const double *uold, const double *vold, const double *pold,
                 double *cu, double *cv, double *z, double *h,
                 double *unew, double *vnew, double *pnew)
{
    int i, j;
    /* intermediates on the box grown towards their consumers */
    for (j = ys; j <= ye + 1; j++)
        for (i = xs - 1; i <= xe; i++) compute_cu_code(i, j, cu, p, u, ld);
    for (j = ys - 1; j <= ye; j++)
        for (i = xs; i <= xe + 1; i++) compute_cv_code(i, j, cv, p, v, ld);
    for (j = ys - 1; j <= ye; j++)
        for (i = xs - 1; i <= xe; i++) compute_z_code(i, j, z, p, u, v, q->fsdx, q->fsdy, ld);
    for (j = ys; j <= ye + 1; j++)
        for (i = xs; i <= xe + 1; i++) compute_h_code(i, j, h, p, u, v, ld);
    for (j = ys; j <= ye; j++)
        for (i = xs; i <= xe; i++) compute_unew_code(i, j, unew, uold, z, cv, h, q->tdts8, q->tdtsdx, ld);
    for (j = ys; j <= ye; j++)
        for (i = xs; i <= xe; i++) compute_vnew_code(i, j, vnew, vold, z, cu, h, q->tdts8, q->tdtsdy, ld);
    for (j = ys; j <= ye; j++)
        for (i = xs; i <= xe; i++) compute_pnew_code(i, j, pnew, pold, cu, cv, q->tdtsdx, q->tdtsdy, ld);
}

/* ------------------------------------------------------------------------ */
/* Shallow-water kernels, SW offset -- the staggering of the GOcean `shallow` benchmark the SURVEY
 * (section 8 f.2) names [external; not in /root/reference]: u(i,j) is the WEST face of T(i,j),
 * v(i,j) the south face, z(i,j) the SW corner.  Formulas frozen in DESIGN.md section 6.2. */
static inline void compute_cu_sw_code(int i, int j, double *cu, const double *p, const double *u, int ld)
{
    A(cu, i, j) = 0.5 * (A(p, i, j) + A(p, i - 1, j)) * A(u, i, j);
}
static inline void compute_cv_sw_code(int i, int j, double *cv, const double *p, const double *v, int ld)
{
    A(cv, i, j) = 0.5 * (A(p, i, j) + A(p, i, j - 1)) * A(v, i, j);
}
static inline void compute_z_sw_code(int i, int j, double *z, const double *p, const double *u,
                                     const double *v, double fsdx, double fsdy, int ld)
{
    A(z, i, j) = (fsdx * (A(v, i, j) - A(v, i - 1, j)) - fsdy * (A(u, i, j) - A(u, i, j - 1))) /
                 (A(p, i - 1, j - 1) + A(p, i, j - 1) + A(p, i, j) + A(p, i - 1, j));
}
static inline void compute_h_sw_code(int i, int j, double *h, const double *p, const double *u,
                                     const double *v, int ld)
{
    A(h, i, j) = A(p, i, j) + 0.25 * (A(u, i + 1, j) * A(u, i + 1, j) + A(u, i, j) * A(u, i, j) +
                                      A(v, i, j + 1) * A(v, i, j + 1) + A(v, i, j) * A(v, i, j));
}
static inline void compute_unew_sw_code(int i, int j, double *unew, const double *uold, const double *z,
                                        const double *cv, const double *h, double tdts8, double tdtsdx, int ld)
{
    A(unew, i, j) = A(uold, i, j) +
                    tdts8 * (A(z, i, j + 1) + A(z, i, j)) *
                        (A(cv, i, j + 1) + A(cv, i - 1, j + 1) + A(cv, i - 1, j) + A(cv, i, j)) -
                    tdtsdx * (A(h, i, j) - A(h, i - 1, j));
}
static inline void compute_vnew_sw_code(int i, int j, double *vnew, const double *vold, const double *z,
                                        const double *cu, const double *h, double tdts8, double tdtsdy, int ld)
{
    A(vnew, i, j) = A(vold, i, j) -
                    tdts8 * (A(z, i + 1, j) + A(z, i, j)) *
                        (A(cu, i + 1, j) + A(cu, i, j) + A(cu, i, j - 1) + A(cu, i + 1, j - 1)) -
                    tdtsdy * (A(h, i, j) - A(h, i, j - 1));
}
static inline void compute_pnew_sw_code(int i, int j, double *pnew, const double *pold, const double *cu,
                                        const double *cv, double tdtsdx, double tdtsdy, int ld)
{
    A(pnew, i, j) = A(pold, i, j) - tdtsdx * (A(cu, i + 1, j) - A(cu, i, j)) -
                    tdtsdy * (A(cv, i, j + 1) - A(cv, i, j));
}

void orc_sw_step_sw(const orc_sw_params *q, int ld, int xs, int xe, int ys, int ye,
                    const double *u, const double *v, const double *p,
                    const double *uold, const double *vold, const double *pold,
                    double *cu, double *cv, double *z, double *h,
                    double *unew, double *vnew, double *pnew)
{
    int i, j;
    /* intermediates on the box grown towards their consumers */
    for (j = ys - 1; j <= ye; j++)
        for (i = xs; i <= xe + 1; i++) compute_cu_sw_code(i, j, cu, p, u, ld);
    for (j = ys; j <= ye + 1; j++)
        for (i = xs - 1; i <= xe; i++) compute_cv_sw_code(i, j, cv, p, v, ld);
    for (j = ys; j <= ye + 1; j++)
        for (i = xs; i <= xe + 1; i++) compute_z_sw_code(i, j, z, p, u, v, q->fsdx, q->fsdy, ld);
    for (j = ys - 1; j <= ye; j++)
        for (i = xs - 1; i <= xe; i++) compute_h_sw_code(i, j, h, p, u, v, ld);
    for (j = ys; j <= ye; j++)
        for (i = xs; i <= xe; i++) compute_unew_sw_code(i, j, unew, uold, z, cv, h, q->tdts8, q->tdtsdx, ld);
    for (j = ys; j <= ye; j++)
        for (i = xs; i <= xe; i++) compute_vnew_sw_code(i, j, vnew, vold, z, cu, h, q->tdts8, q->tdtsdy, ld);
    for (j = ys; j <= ye; j++)
        for (i = xs; i <= xe; i++) compute_pnew_sw_code(i, j, pnew, pold, cu, cv, q->tdtsdx, q->tdtsdy, ld);
}

/* time_smooth of the GOcean `shallow` benchmark [external; not in /root/reference] -- the Asselin filter of
 * the leapfrog scheme, pointwise, formula frozen in DESIGN.md section 6.3 */
static inline void time_smooth_code(int i, int j, const double *field, const double *field_new, double *field_old,
                                    double alpha, int ld)
{
    A(field_old, i, j) = A(field, i, j) + alpha * (A(field_new, i, j) - 2.0 * A(field, i, j) + A(field_old, i, j));
}

/* ONE kernel of the set over a 1-based inclusive box: the PSy loop nest `do j / do i / call kern_code(i, j, ...)`
 * (form: infrastructure_mod.f90:32-41) -- the checker of the per-kernel launch entries dlesm_compute_*_f64.
 * kernel: 0 cu(out; p, u)  1 cv(out; p, v)  2 z(out; p, u, v; s0 = fsdx, s1 = fsdy)  3 h(out; p, u, v)
 *         4 unew(out; uold, z, cv, h; tdts8, tdtsdx)  5 vnew(out; vold, z, cu, h; tdts8, tdtsdy)
 *         6 pnew(out; pold, cu, cv; tdtsdx, tdtsdy)   7 time_smooth(out = field_old; field, field_new, field_old; alpha)
 * sw_offset: 0 = NE staggering, 1 = SW staggering.  Returns -1 for an unknown kernel. */
int orc_sw_kernel(int kernel, int sw_offset, int ld, int xs, int xe, int ys, int ye, double s0, double s1,
                  double *out, const double *a, const double *b, const double *c, const double *d)
{
    int i, j;
    if (kernel < 0 || kernel > 7) return -1;
    for (j = ys; j <= ye; j++)
        for (i = xs; i <= xe; i++) {
            if (!sw_offset) switch (kernel) {
                case 0: compute_cu_code(i, j, out, a, b, ld); break;
                case 1: compute_cv_code(i, j, out, a, b, ld); break;
                case 2: compute_z_code(i, j, out, a, b, c, s0, s1, ld); break;
                case 3: compute_h_code(i, j, out, a, b, c, ld); break;
                case 4: compute_unew_code(i, j, out, a, b, c, d, s0, s1, ld); break;
                case 5: compute_vnew_code(i, j, out, a, b, c, d, s0, s1, ld); break;
                case 6: compute_pnew_code(i, j, out, a, b, c, s0, s1, ld); break;
                default: time_smooth_code(i, j, a, b, out, s0, ld); break;
            } else switch (kernel) {
                case 0: compute_cu_sw_code(i, j, out, a, b, ld); break;
                case 1: compute_cv_sw_code(i, j, out, a, b, ld); break;
                case 2: compute_z_sw_code(i, j, out, a, b, c, s0, s1, ld); break;
                case 3: compute_h_sw_code(i, j, out, a, b, c, ld); break;
                case 4: compute_unew_sw_code(i, j, out, a, b, c, d, s0, s1, ld); break;
                case 5: compute_vnew_sw_code(i, j, out, a, b, c, d, s0, s1, ld); break;
                case 6: compute_pnew_sw_code(i, j, out, a, b, c, s0, s1, ld); break;
                default: time_smooth_code(i, j, a, b, out, s0, ld); break;
            }
        }
    return 0;
}

/* init_periodic_bc_halos, field_mod.f90:1394-1464: source/dest hold up to 4 regions each, in the
 * reference's order; returns their number.  bc: 0 = GO_BC_PERIODIC (grid_mod.f90:64-69). */
int orc_periodic_halos(const orc_region *it, int bc_x, int bc_y, orc_region *source, orc_region *dest)
{
    int n = 0;
    if (bc_x == 0) {
        /* E-most column set to W-most internal column (:1413-1423) */
        dest[n].xstart = it->xstop + 1;  dest[n].xstop = it->xstop + 1;
        dest[n].ystart = it->ystart;     dest[n].ystop = it->ystop;
        source[n].xstart = it->xstart;   source[n].xstop = it->xstart;
        source[n].ystart = it->ystart;   source[n].ystop = it->ystop;
        n++;
        /* W-most column set to E-most internal column (:1425-1435) */
        dest[n].xstart = it->xstart - 1; dest[n].xstop = it->xstart - 1;
        dest[n].ystart = it->ystart;     dest[n].ystop = it->ystop;
        source[n].xstart = it->xstop;    source[n].xstop = it->xstop;
        source[n].ystart = it->ystart;   source[n].ystop = it->ystop;
        n++;
    }
    if (bc_y == 0) {
        /* N-most row set to S-most internal row (:1439-1449) */
        dest[n].xstart = it->xstart - 1; dest[n].xstop = it->xstop + 1;
        dest[n].ystart = it->ystop + 1;  dest[n].ystop = it->ystop + 1;
        source[n].xstart = it->xstart - 1; source[n].xstop = it->xstop + 1;
        source[n].ystart = it->ystart;   source[n].ystop = it->ystart;
        n++;
        /* S-most row set to N-most internal row (:1451-1461) */
        dest[n].xstart = it->xstart - 1; dest[n].xstop = it->xstop + 1;
        dest[n].ystart = it->ystart - 1; dest[n].ystop = it->ystart - 1;
        source[n].xstart = it->xstart - 1; source[n].xstop = it->xstop + 1;
        source[n].ystart = it->ystop;    source[n].ystop = it->ystop;
        n++;
    }
    for (int k = 0; k < n; k++) {
        source[k].nx = source[k].xstop - source[k].xstart + 1;  source[k].ny = source[k].ystop - source[k].ystart + 1;
        dest[k].nx = dest[k].xstop - dest[k].xstart + 1;        dest[k].ny = dest[k].ystop - dest[k].ystart + 1;
    }
    return n;
}

/* the copies themselves, copy_2dfield_patch (field_mod.f90:1179-1187), in the list's order */
void orc_apply_periodic_halos(double *f, int ld, const orc_region *it, int bc_x, int bc_y)
{
    orc_region src[4], dst[4];
    const int n = orc_periodic_halos(it, bc_x, bc_y, src, dst);
    for (int k = 0; k < n; k++)
        for (int j = 0; j < src[k].ny; j++)
            for (int i = 0; i < src[k].nx; i++)
                f[IDX(ld, dst[k].xstart + i, dst[k].ystart + j)] = f[IDX(ld, src[k].xstart + i, src[k].ystart + j)];
}
