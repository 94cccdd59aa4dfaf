/* comd_oracle.c -- TEST INFRASTRUCTURE ONLY (see comd_oracle.h).
 *
 * A CPU restatement of the reference's hot path.  Every function names the reference
 * file:line (under /root/reference/src-mpi) whose behaviour it restates.  The code is
 * written from the algorithm, not copied: one process simulates all ranks of the
 * Cartesian decomposition ("virtual ranks"), the force loops are the full 27-cell
 * stencil form the reference's device kernels use (gpu_lj_thread_atom.h:29-143,
 * gpu_eam_thread_atom.h:32-140) rather than the Verlet-list form of its CPU path
 * (ljForce.c:146-265, eam.c:266-419) -- the two differ only in summation order, which
 * the reference itself shows to be invisible at 12 digits.
 */
#include "comd_oracle.h"

/* mytype.h:8-21: the arithmetic type of the run.  -DORACLE_SINGLE restates the reference's single-precision build (real_t = float; its
 * C sources keep double literals, so mixed expressions are evaluated in double and rounded on assignment -- the same happens here).
 * The C interface (comd_oracle.h) stays in double either way. */
#ifdef ORACLE_SINGLE
typedef float real_t;
#define RFMT "%g"
#define REFMT "%e"
#else
typedef double real_t;
#define RFMT "%lg"
#define REFMT "%le"
#endif

#include <assert.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* ---- constants.h:14-39 ------------------------------------------------------------ */
#define AMU_KG      1.660538921e-27
#define FS_S        1.0e-15
#define ANG_M       1.0e-10
#define EV_J        1.602176565e-19
static const double kAmuToInternalMass = AMU_KG * ANG_M * ANG_M / (FS_S * FS_S * EV_J);
static const double kB_eV        = 8.6173324e-5;
static const double kHartreeToEv = 27.21138505;
static const double kBohrToAngs  = 0.52917721092;

/* ---- random.c:21-75 ---------------------------------------------------------------- */
double oracle_lcg61(uint64_t* seed)
{
   const uint64_t modulus = UINT64_C(2305843009213693951);           /* 2^61 - 1 */
   const double   toUnit  = 1.0 / UINT64_C(2305843009213693951);      /* random.c:26: double in either build */
   *seed = (*seed * UINT64_C(437799614237992725)) % modulus;          /* wraps mod 2^64 first, as the reference does */
   return *seed * toUnit;
}

uint64_t oracle_mkSeed(uint32_t id, uint32_t callSite)
{
   const uint32_t knuth = UINT32_C(2654435761);
   uint32_t hi = id * knuth;
   uint32_t lo = (id + callSite) * knuth;
   uint64_t seed = (UINT64_C(0x100000000) * hi) + lo;
   for (int k = 0; k < 10; ++k) oracle_lcg61(&seed);
   return seed;
}

double oracle_gasdev(uint64_t* seed)
{
   real_t a, b, s;
   do {
      a = 2.0 * oracle_lcg61(seed) - 1.0;
      b = 2.0 * oracle_lcg61(seed) - 1.0;
      s = a * a + b * b;
   } while (s >= 1.0 || s == 0.0);
   const real_t g = b * sqrt(-2.0 * log(s) / s);                        /* random.c:72 returns real_t: rounded to the build's precision */
   return g;
}

/* ---- data ---------------------------------------------------------------------------- */
typedef struct {            /* CoMDTypes.h:169-176 + eam.c:496-519 (values[-1..n+1]) */
   int n; real_t x0, invDx; real_t* v;   /* v points at values[0]; v[-1], v[n], v[n+1] valid */
} Table;

typedef struct { int gid, type; real_t rx, ry, rz, px, py, pz; } AtomRec;   /* haloExchange.h:32-38 */

typedef struct {
   /* decomposition.c:18-50 */
   int    coord[3];
   real_t lmin[3], lmax[3], lext[3];
   /* linkCells.c:122-182 */
   int    g[3], nLocal, nTotal;
   real_t bsize[3], binv[3];
   int*   nAtoms;
   /* initAtoms.c:26-60 */
   int    *gid, *spec;
   real_t *r[3], *p[3], *f[3], *U, *rhobar, *dfE;
   /* haloExchange.c:198-328, 345-475 */
   int    nbr[6];
   real_t shift[6][3];
   int    nAtomCells[6], *atomCells[6];
   int    nForceCells[6], *fSend[6], *fRecv[6];
   int*   nbrBoxes;          /* [nLocal*27], self first (gpu_utility.c:520-531) */
   AtomRec *sendBuf[2], *recvBuf[2]; int nSend[2], nRecv[2];
   real_t  *fsend[2], *frecv[2];   int nfSend[2], nfRecv[2];
   real_t ePot, eKin;
} Rank;

struct OracleSim {
   int nx, ny, nz, pg[3], nRanks, cap, doeam, nGlobal;
   real_t lat, dt, gmin[3], gmax[3], gext[3];
   real_t cutoff, mass;
   /* LJ (ljForce.c:102-120) */
   real_t sigma, epsilon;
   /* EAM */
   Table phi, rho, F;
   real_t *phiSpline, *rhoSpline;      /* -P (oracle_use_splines): 4 n coefficients each, else NULL */
   Rank* rk;
   real_t ePot, eKin; double loopSeconds;
};

static double wallSeconds(void)
{
   struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts);
   return ts.tv_sec + 1e-9 * ts.tv_nsec;
}

/* ---- linkCells.c:299-346 getBoxFromTuple ------------------------------------------- */
static int boxFromTuple(const Rank* k, int ix, int iy, int iz)
{
   const int gx = k->g[0], gy = k->g[1], gz = k->g[2];
   const int base = k->nLocal;
   if (iz == gz)  return base + 2*gz*gy + 2*gz*(gx+2) + (gx+2)*(gy+2) + (gx+2)*(iy+1) + (ix+1);
   if (iz == -1)  return base + 2*gz*gy + 2*gz*(gx+2) + (gx+2)*(iy+1) + (ix+1);
   if (iy == gy)  return base + 2*gz*gy + gz*(gx+2) + (gx+2)*iz + (ix+1);
   if (iy == -1)  return base + 2*gz*gy + iz*(gx+2) + (ix+1);
   if (ix == gx)  return base + gy*gz + iz*gy + iy;
   if (ix == -1)  return base + iz*gy + iy;
   return ix + gx*(iy + gy*iz);     /* IDX3D, no Hilbert order (out of scope) */
}

/* ---- linkCells.c:448-480 getBoxFromCoord (tie rules included) ------------------------ */
static int boxFromCoord(const Rank* k, const real_t r[3])
{
   int t[3];
   for (int a = 0; a < 3; ++a) {
      t[a] = (int)floor((r[a] - k->lmin[a]) * k->binv[a]);
      if (t[a] < -1 || t[a] > k->g[a]) {     /* moved more than one cell in a step: outside the halo, lost */
         fprintf(stderr, "oracle: an atom left the halo region (axis %d, cell %d of %d)\n", a, t[a], k->g[a]); abort();
      }
      if (r[a] < k->lmax[a]) { if (t[a] == k->g[a]) t[a] = k->g[a] - 1; }
      else t[a] = k->g[a];
   }
   return boxFromTuple(k, t[0], t[1], t[2]);
}

static void localTuple(const Rank* k, int iBox, int* ix, int* iy, int* iz)
{
   *ix = iBox % k->g[0]; *iy = (iBox / k->g[0]) % k->g[1]; *iz = iBox / (k->g[0]*k->g[1]);
}

/* ---- linkCells.c:216-248 putAtomInBox ------------------------------------------------- */
static int putAtom(OracleSim* s, Rank* k, int gid, int type, const real_t r[3], const real_t p[3])
{
   int b = boxFromCoord(k, r);
   if (k->nAtoms[b] >= s->cap) {
      fprintf(stderr, "oracle: link cell %d overflows capacity %d\n", b, s->cap); abort();
   }
   int o = b * s->cap + k->nAtoms[b]++;
   k->gid[o] = gid; k->spec[o] = type;
   for (int a = 0; a < 3; ++a) { k->r[a][o] = r[a]; k->p[a][o] = p[a]; }
   return o;
}

static void copySlot(Rank* k, int from, int to)
{
   k->gid[to] = k->gid[from]; k->spec[to] = k->spec[from];
   for (int a = 0; a < 3; ++a) { k->r[a][to] = k->r[a][from]; k->p[a][to] = k->p[a][from]; k->f[a][to] = k->f[a][from]; }
   k->U[to] = k->U[from];
}

/* ---- decomposition.c:57-66 processorNum ---------------------------------------------- */
static int rankAt(const OracleSim* s, const Rank* k, int dx, int dy, int dz)
{
   int c[3] = { (k->coord[0] + dx + s->pg[0]) % s->pg[0],
                (k->coord[1] + dy + s->pg[1]) % s->pg[1],
                (k->coord[2] + dz + s->pg[2]) % s->pg[2] };
   return c[0] + s->pg[0] * (c[1] + s->pg[1] * c[2]);
}

/* ---- haloExchange.c:1543-1567 mkAtomCellList; :1712-1801 mkForceSend/RecvCellList ------ */
static int* cellBlock(const Rank* k, const int lo[3], const int hi[3], int* count)
{
   int n = (hi[0]-lo[0]) * (hi[1]-lo[1]) * (hi[2]-lo[2]);
   int* list = (int*)malloc((size_t)n * sizeof(int));
   int c = 0;
   for (int ix = lo[0]; ix < hi[0]; ++ix)
      for (int iy = lo[1]; iy < hi[1]; ++iy)
         for (int iz = lo[2]; iz < hi[2]; ++iz)
            list[c++] = boxFromTuple(k, ix, iy, iz);
   *count = n;
   return list;
}

static void buildHaloLists(Rank* k)
{
   for (int face = 0; face < 6; ++face) {
      int axis = face / 2, plus = face & 1;
      int lo[3] = { -1, -1, -1 }, hi[3] = { k->g[0]+1, k->g[1]+1, k->g[2]+1 };
      /* atom exchange: the halo plane and the first local plane of that face, full extent in the other axes */
      if (plus) lo[axis] = hi[axis] - 2; else hi[axis] = lo[axis] + 2;
      k->atomCells[face] = cellBlock(k, lo, hi, &k->nAtomCells[face]);

      /* force exchange: footprint grows x -> y -> z */
      int slo[3], shi[3], rlo[3], rhi[3];
      for (int a = 0; a < 3; ++a) {
         if (a < axis)      { slo[a] = -1; shi[a] = k->g[a] + 1; }   /* axes already exchanged: include their halos */
         else               { slo[a] = 0;  shi[a] = k->g[a]; }
         rlo[a] = slo[a]; rhi[a] = shi[a];
      }
      if (plus) { slo[axis] = k->g[axis]-1; shi[axis] = k->g[axis];   rlo[axis] = k->g[axis]; rhi[axis] = k->g[axis]+1; }
      else      { slo[axis] = 0;            shi[axis] = 1;            rlo[axis] = -1;         rhi[axis] = 0; }
      int nr;
      k->fSend[face] = cellBlock(k, slo, shi, &k->nForceCells[face]);
      k->fRecv[face] = cellBlock(k, rlo, rhi, &nr);
      assert(nr == k->nForceCells[face]);
   }
}

/* ---- eam.c:496-519 initInterpolationObject, :557-579 interpolate ----------------------- */
static void tableInit(Table* t, int n, real_t x0, real_t dx, const real_t* data)
{
   real_t* raw = (real_t*)calloc((size_t)n + 3, sizeof(real_t));
   t->v = raw + 1; t->n = n; t->x0 = x0; t->invDx = 1.0 / dx;
   for (int i = 0; i < n; ++i) t->v[i] = data[i];
   t->v[-1] = t->v[0];
   t->v[n] = t->v[n-1]; t->v[n+1] = t->v[n-1];
}

static inline void tableEval(const Table* t, real_t x, real_t* f, real_t* df)
{
   const real_t* v = t->v;
   if (x < t->x0) x = t->x0;
   x = (x - t->x0) * t->invDx;
   int i = (int)floor(x);
   if (i > t->n) { i = t->n; x = t->n / t->invDx; }
   x = x - floor(x);
   real_t g1 = v[i+1] - v[i-1];
   real_t g2 = v[i+2] - v[i];
   *f  = v[i] + 0.5 * x * (g1 + x * (v[i+1] + v[i-1] - 2.0 * v[i]));
   *df = 0.5 * (g1 + x * (g2 - g1)) * t->invDx;
}

/* ---- eam.c:802-872 eamReadFuncfl -------------------------------------------------------- */
static int readFuncfl(OracleSim* s, const char* dir, const char* name)
{
   char path[4096], line[4096];
   snprintf(path, sizeof path, "%s/%s", dir, name);
   FILE* fp = fopen(path, "r");
   if (!fp) { fprintf(stderr, "oracle: cannot open %s\n", path); return -1; }
   if (!fgets(line, sizeof line, fp)) { fclose(fp); return -1; }                /* comment / element */
   int z; double amu, lat; char ltype[16];
   if (!fgets(line, sizeof line, fp) || sscanf(line, "%d %le %le %15s", &z, &amu, &lat, ltype) != 4) { fclose(fp); return -1; }
   int nRho, nR; double dRho, dR, rc;
   if (!fgets(line, sizeof line, fp) || sscanf(line, "%d %le %d %le %le", &nRho, &dRho, &nR, &dR, &rc) != 5) { fclose(fp); return -1; }
   s->lat = lat; s->mass = amu * kAmuToInternalMass; s->cutoff = rc;
   int nb = nRho > nR ? nRho : nR;
   real_t* buf = (real_t*)malloc((size_t)nb * sizeof(real_t));
   for (int i = 0; i < nRho; ++i) if (fscanf(fp, RFMT, buf + i) != 1) { fclose(fp); free(buf); return -1; }
   tableInit(&s->F, nRho, 0.0, dRho, buf);
   for (int i = 0; i < nR; ++i) if (fscanf(fp, RFMT, buf + i) != 1) { fclose(fp); free(buf); return -1; }
   for (int i = 1; i < nR; ++i) {            /* Z(r) -> phi(r) = Z^2/r in eV */
      real_t r = 0.0 + i * dR;
      buf[i] *= buf[i] / r;
      buf[i] *= kHartreeToEv * kBohrToAngs;
   }
   buf[0] = buf[1] + (buf[1] - buf[2]);
   tableInit(&s->phi, nR, 0.0, dR, buf);
   for (int i = 0; i < nR; ++i) if (fscanf(fp, RFMT, buf + i) != 1) { fclose(fp); free(buf); return -1; }
   tableInit(&s->rho, nR, 0.0, dR, buf);
   free(buf); fclose(fp);
   return 0;
}

/* ---- eam.c:680-757 eamReadSetfl (single element) ------------------------------------------ */
static int readSetfl(OracleSim* s, const char* dir, const char* name)
{
   char path[4096], line[4096];
   snprintf(path, sizeof path, "%s/%s", dir, name);
   FILE* fp = fopen(path, "r");
   if (!fp) { fprintf(stderr, "oracle: cannot open %s\n", path); return -1; }
   for (int i = 0; i < 3; ++i) if (!fgets(line, sizeof line, fp)) { fclose(fp); return -1; }      /* comments */
   int nElems = 0;
   if (!fgets(line, sizeof line, fp) || sscanf(line, "%d", &nElems) != 1 || nElems != 1) { fclose(fp); return -1; }
   int nRho, nR; double dRho, dR, rc;
   if (!fgets(line, sizeof line, fp) || sscanf(line, "%d %le %d %le %le", &nRho, &dRho, &nR, &dR, &rc) != 5) { fclose(fp); return -1; }
   int z; double amu, lat; char ltype[16];
   if (!fgets(line, sizeof line, fp) || sscanf(line, "%d %le %le %15s", &z, &amu, &lat, ltype) != 4) { fclose(fp); return -1; }
   s->lat = lat; s->mass = amu * kAmuToInternalMass; s->cutoff = rc;
   int nb = nRho > nR ? nRho : nR;
   real_t* buf = (real_t*)malloc((size_t)nb * sizeof(real_t));
   for (int i = 0; i < nRho; ++i) if (fscanf(fp, RFMT, buf + i) != 1) { fclose(fp); free(buf); return -1; }
   tableInit(&s->F, nRho, 0.0, dRho, buf);
   for (int i = 0; i < nR; ++i) if (fscanf(fp, RFMT, buf + i) != 1) { fclose(fp); free(buf); return -1; }
   tableInit(&s->rho, nR, 0.0, dR, buf);
   for (int i = 0; i < nR; ++i) if (fscanf(fp, RFMT, buf + i) != 1) { fclose(fp); free(buf); return -1; }
   for (int i = 1; i < nR; ++i) buf[i] /= (0.0 + i * dR);          /* the file stores r * phi(r) */
   buf[0] = buf[1] + (buf[1] - buf[2]);
   tableInit(&s->phi, nR, 0.0, dR, buf);
   free(buf); fclose(fp);
   return 0;
}

/* ---- timestep.c:109-133 kineticEnergy (sum over virtual ranks = addRealParallel) -------- */
void oracle_kinetic_energy(OracleSim* s)
{
   real_t eP = 0.0, eK = 0.0;
   for (int ir = 0; ir < s->nRanks; ++ir) {
      Rank* k = &s->rk[ir];
      real_t loc = 0.0;
      const real_t halfInvMass = 0.5 / s->mass;
      for (int b = 0; b < k->nLocal; ++b)
         for (int o = b * s->cap, e = o + k->nAtoms[b]; o < e; ++o)
            loc += (k->p[0][o]*k->p[0][o] + k->p[1][o]*k->p[1][o] + k->p[2][o]*k->p[2][o]) * halfInvMass;
      k->eKin = loc;
      eP += k->ePot; eK += loc;
   }
   s->ePot = eP; s->eKin = eK;
}

/* ---- timestep.c:143-180 advanceVelocityCpu / advancePositionCpu -------------------------- */
void oracle_advance_velocity(OracleSim* s, double dt)
{
   for (int ir = 0; ir < s->nRanks; ++ir) {
      Rank* k = &s->rk[ir];
      for (int b = 0; b < k->nLocal; ++b)
         for (int o = b * s->cap, e = o + k->nAtoms[b]; o < e; ++o)
            for (int a = 0; a < 3; ++a) k->p[a][o] += dt * k->f[a][o];
   }
}

void oracle_advance_position(OracleSim* s, double dt)
{
   const real_t invMass = 1.0 / s->mass;
   for (int ir = 0; ir < s->nRanks; ++ir) {
      Rank* k = &s->rk[ir];
      for (int b = 0; b < k->nLocal; ++b)
         for (int o = b * s->cap, e = o + k->nAtoms[b]; o < e; ++o)
            for (int a = 0; a < 3; ++a) k->r[a][o] += dt * k->p[a][o] * invMass;
   }
}

/* ---- linkCells.c:364-410 updateLinkCellsCpu + moveAtom ------------------------------------ */
static void updateLinkCells(OracleSim* s, Rank* k)
{
   for (int b = k->nLocal; b < k->nTotal; ++b) k->nAtoms[b] = 0;     /* emptyHaloCells */
   for (int b = 0; b < k->nLocal; ++b) {
      int i = 0;
      while (i < k->nAtoms[b]) {
         int o = b * s->cap + i;
         real_t r[3] = { k->r[0][o], k->r[1][o], k->r[2][o] };
         int nb = boxFromCoord(k, r);
         if (nb == b) { ++i; continue; }
         if (k->nAtoms[nb] >= s->cap) { fprintf(stderr, "oracle: cell overflow in updateLinkCells\n"); abort(); }
         copySlot(k, o, nb * s->cap + k->nAtoms[nb]++);
         int last = --k->nAtoms[b];
         if (last != i) copySlot(k, b * s->cap + last, o);
      }
   }
}

static int cmpRec(const void* a, const void* b)
{
   int x = ((const AtomRec*)a)->gid, y = ((const AtomRec*)b)->gid;
   return (x > y) - (x < y);
}

/* The reference keeps boundary and halo cells in gid order so that the force exchange pairs
 * atoms correctly (haloExchange.c:1918-1965 sortAtomsInCell, gpu_kernels.cu:1013-1043).
 * The oracle keeps EVERY cell in gid order: cell membership then fixes the order. */
static void sortCells(OracleSim* s, Rank* k)
{
   AtomRec* tmp = (AtomRec*)malloc((size_t)s->cap * sizeof(AtomRec));
   for (int b = 0; b < k->nTotal; ++b) {
      int n = k->nAtoms[b], base = b * s->cap, sorted = 1;
      for (int i = 1; i < n; ++i) if (k->gid[base+i-1] > k->gid[base+i]) { sorted = 0; break; }
      if (sorted) continue;
      for (int i = 0; i < n; ++i) {
         int o = base + i;
         tmp[i] = (AtomRec){ k->gid[o], k->spec[o], k->r[0][o], k->r[1][o], k->r[2][o], k->p[0][o], k->p[1][o], k->p[2][o] };
      }
      qsort(tmp, (size_t)n, sizeof(AtomRec), cmpRec);
      for (int i = 0; i < n; ++i) {
         int o = base + i;
         k->gid[o] = tmp[i].gid; k->spec[o] = tmp[i].type;
         k->r[0][o] = tmp[i].rx; k->r[1][o] = tmp[i].ry; k->r[2][o] = tmp[i].rz;
         k->p[0][o] = tmp[i].px; k->p[1][o] = tmp[i].py; k->p[2][o] = tmp[i].pz;
      }
   }
   free(tmp);
}

/* ---- haloExchange.c:1576-1647 loadAtomsBuffer (CPU branch) -------------------------------- */
static int packAtoms(const OracleSim* s, const Rank* k, int face, AtomRec* buf)
{
   int n = 0;
   const real_t* sh = k->shift[face];
   for (int c = 0; c < k->nAtomCells[face]; ++c) {
      int b = k->atomCells[face][c];
      for (int o = b * s->cap, e = o + k->nAtoms[b]; o < e; ++o)
         buf[n++] = (AtomRec){ k->gid[o], k->spec[o],
                               k->r[0][o] + sh[0], k->r[1][o] + sh[1], k->r[2][o] + sh[2],
                               k->p[0][o], k->p[1][o], k->p[2][o] };
   }
   return n;
}

/* ---- haloExchange.c:1493-1522 exchangeData, for every virtual rank at once ---------------- */
static void exchangeAtoms(OracleSim* s)
{
   for (int axis = 0; axis < 3; ++axis) {
      int fm = 2*axis, fp = fm + 1;
      for (int ir = 0; ir < s->nRanks; ++ir) {
         Rank* k = &s->rk[ir];
         k->nSend[0] = packAtoms(s, k, fm, k->sendBuf[0]);
         k->nSend[1] = packAtoms(s, k, fp, k->sendBuf[1]);
      }
      /* my minus-face message lands in my minus neighbour's "from plus" buffer, and vice versa */
      for (int ir = 0; ir < s->nRanks; ++ir) {
         Rank* k = &s->rk[ir];
         Rank* m = &s->rk[k->nbr[fm]];
         Rank* p = &s->rk[k->nbr[fp]];
         memcpy(m->recvBuf[1], k->sendBuf[0], (size_t)k->nSend[0] * sizeof(AtomRec)); m->nRecv[1] = k->nSend[0];
         memcpy(p->recvBuf[0], k->sendBuf[1], (size_t)k->nSend[1] * sizeof(AtomRec)); p->nRecv[0] = k->nSend[1];
      }
      for (int ir = 0; ir < s->nRanks; ++ir) {
         Rank* k = &s->rk[ir];
         for (int side = 0; side < 2; ++side)          /* unload faceM (from minus nbr) then faceP */
            for (int i = 0; i < k->nRecv[side]; ++i) {
               const AtomRec* a = &k->recvBuf[side][i];
               real_t r[3] = { a->rx, a->ry, a->rz }, p[3] = { a->px, a->py, a->pz };
               putAtom(s, k, a->gid, a->type, r, p);
            }
      }
   }
}

/* ---- timestep.c:222-276 redistributeAtomsGpu / :356-405 redistributeAtomsCpuNL ------------- */
void oracle_redistribute(OracleSim* s)
{
   for (int ir = 0; ir < s->nRanks; ++ir) updateLinkCells(s, &s->rk[ir]);
   exchangeAtoms(s);
   for (int ir = 0; ir < s->nRanks; ++ir) sortCells(s, &s->rk[ir]);
}

/* ---- LJ: ljForce.c:146-265 maths, stencil form of gpu_lj_thread_atom.h:29-143 -------------- */
static void ljForceRank(const OracleSim* s, Rank* k)
{
   const real_t rc2 = s->cutoff * s->cutoff;
   const real_t s6 = s->sigma*s->sigma*s->sigma*s->sigma*s->sigma*s->sigma;
   const real_t rc6 = s6 / (rc2*rc2*rc2);
   const real_t eShift = 1.0 * rc6 * (rc6 - 1.0);                 /* POT_SHIFT 1.0 (ljForce.c:83) */
   const real_t eps = s->epsilon;
   const int cap = s->cap;
   real_t ePot = 0.0;
#pragma omp parallel for schedule(dynamic, 4) reduction(+:ePot)
   for (int b = 0; b < k->nLocal; ++b) {
      const int* nb = &k->nbrBoxes[b * 27];
      for (int io = b * cap, ie = io + k->nAtoms[b]; io < ie; ++io) {
         const real_t xi = k->r[0][io], yi = k->r[1][io], zi = k->r[2][io];
         real_t fx = 0.0, fy = 0.0, fz = 0.0, e = 0.0;
         for (int q = 0; q < 27; ++q) {
            const int jb = nb[q];
            for (int jo = jb * cap, je = jo + k->nAtoms[jb]; jo < je; ++jo) {
               real_t dx = xi - k->r[0][jo], dy = yi - k->r[1][jo], dz = zi - k->r[2][jo];
               real_t r2 = dx*dx + dy*dy + dz*dz;
               if (r2 <= rc2 && r2 > 0.0) {
                  real_t ir2 = 1.0 / r2;
                  real_t r6 = s6 * (ir2*ir2*ir2);
                  e += 0.5 * (r6 * (r6 - 1.0) - eShift);
                  real_t fr = r6 * ir2 * (48.0 * r6 - 24.0);
                  fx += fr * dx; fy += fr * dy; fz += fr * dz;
               }
            }
         }
         k->f[0][io] = fx * eps; k->f[1][io] = fy * eps; k->f[2][io] = fz * eps;
         k->U[io] = e * 4.0 * eps;
         ePot += k->U[io];
      }
   }
   k->ePot = ePot;
}

/* ---- -P: cubic splines in x = r^2 (gpu_utility.c:377-430 initSplineCoefficients, gpu_common.h:95-129 interpolateSpline) --------
 * The reference has this mode on the GPU only (its CPU eamForce always interpolates quadratically), so these two functions restate
 * device code and the results they check are PARITY-UNPINNED against reference output: none exists for -P. */
static real_t* splineCoefficients(const Table* t)
{
   const int n = t->n; const real_t x0 = t->x0, invDx = t->invDx; const real_t* v = t->v;
   real_t* u = (real_t*)malloc((size_t)n * sizeof(real_t));
   real_t* y2 = (real_t*)malloc((size_t)(n + 1) * sizeof(real_t));
   y2[0] = 0.0; u[0] = 0.0;                                  /* second derivative 0 at the first knot */
   for (int i = 1; i < n; ++i) {
      const real_t xi = (x0 + i / invDx) * (x0 + i / invDx), xp = (x0 + (i - 1) / invDx) * (x0 + (i - 1) / invDx), xn = (x0 + (i + 1) / invDx) * (x0 + (i + 1) / invDx);
      const real_t sig = (xi - xp) / (xn - xp), p = sig * y2[i - 1] + 2.0;
      y2[i] = (sig - 1.0) / p;
      u[i] = (v[i + 1] - v[i]) / (xn - xi) - (v[i] - v[i - 1]) / (xi - xp);
      u[i] = (6.0 * u[i] / (xn - xp) - sig * u[i - 1]) / p;
   }
   const real_t xN = (x0 + n / invDx) * (x0 + n / invDx), xNp = (x0 + (n - 1) / invDx) * (x0 + (n - 1) / invDx);
   const real_t qn = 0.5, un = (-3.0 / (xN - xNp)) * (v[n] - v[n - 1]) / (xN - xNp);      /* first derivative 0 at the last knot */
   y2[n] = (un - qn * u[n - 1]) / (qn * y2[n - 1] + 1.0);
   for (int i = n - 1; i >= 0; --i) y2[i] = y2[i] * y2[i + 1] + u[i];
   real_t* c = (real_t*)malloc((size_t)4 * n * sizeof(real_t));
   for (int i = 0; i < n; ++i) {
      const real_t x1 = (x0 + i / invDx) * (x0 + i / invDx), x2 = (x0 + (i + 1) / invDx) * (x0 + (i + 1) / invDx);
      const real_t d1 = y2[i], d2 = y2[i + 1], y1 = v[i], yy2 = v[i + 1];
      c[4*i]   = 1.0 / (6.0 * (x2 - x1)) * (d2 - d1);
      c[4*i+1] = 1.0 / (2.0 * (x2 - x1)) * (x2 * d1 - x1 * d2);
      c[4*i+2] = 1.0 / (x2 - x1) * (1.0/6.0 * (-3*x2*x2 + (x2-x1)*(x2-x1)) * d1 + 1.0/6.0 * (3*x1*x1 - (x2-x1)*(x2-x1)) * d2 - y1 + yy2);
      c[4*i+3] = 1 / (x2 - x1) * (x2*y1 - x1*yy2 + 1.0/6.0 * d1 * (x2*x2*x2 - x2*(x2-x1)*(x2-x1)) + 1.0/6.0 * d2 * (-x1*x1*x1 + x1*(x2-x1)*(x2-x1)));
   }
   free(u); free(y2);
   return c;
}

/* value and (1/r) d/dr at r^2; the interval is picked in single precision like the device code */
static inline void splineEval(const Table* t, const real_t* c, real_t r2, real_t* f, real_t* df)
{
   float r = sqrtf((float)r2);
   const float x0 = (float)t->x0, xn = (float)(t->x0 + t->n / t->invDx), invDx = (float)t->invDx, invDxXx0 = (float)(t->invDx * t->x0);
   r = r > x0 ? r : x0; r = r < xn ? r : xn;
   r = r * invDx - invDxXx0;
   int ii = (int)floorf(r);
   if (ii > t->n - 1) ii = t->n - 1;
   const real_t a = c[4*ii], b = c[4*ii+1], cc = c[4*ii+2], d = c[4*ii+3];
   const real_t tmp = a * r2 + b;
   *f = (tmp * r2 + cc) * r2 + d;
   *df = 2.0 * ((3.0 * tmp - b) * r2 + cc);
}

/* ---- EAM pass 1 and 3: eam.c:266-419 maths, stencil form of gpu_eam_thread_atom.h:32-140 ---- */
static void eamPass(const OracleSim* s, Rank* k, int pass)
{
   const real_t rc2 = s->cutoff * s->cutoff;
   const int cap = s->cap;
#pragma omp parallel for schedule(dynamic, 16)
   for (int b = 0; b < k->nLocal; ++b) {
      const int* nb = &k->nbrBoxes[b * 27];
      for (int io = b * cap, ie = io + k->nAtoms[b]; io < ie; ++io) {
         const real_t xi = k->r[0][io], yi = k->r[1][io], zi = k->r[2][io];
         real_t fx = 0.0, fy = 0.0, fz = 0.0, e = 0.0, rb = 0.0;
         if (pass == 3) { fx = k->f[0][io]; fy = k->f[1][io]; fz = k->f[2][io]; }
         for (int q = 0; q < 27; ++q) {
            const int jb = nb[q];
            for (int jo = jb * cap, je = jo + k->nAtoms[jb]; jo < je; ++jo) {
               real_t dx = xi - k->r[0][jo], dy = yi - k->r[1][jo], dz = zi - k->r[2][jo];
               real_t r2 = dx*dx + dy*dy + dz*dz;
               if (r2 <= rc2 && r2 > 0.0 && s->phiSpline) {             /* -P: gpu_eam_thread_atom.h:104-121 */
                  real_t phi, dphi, rho, drho;
                  splineEval(&s->rho, s->rhoSpline, r2, &rho, &drho);
                  if (pass == 1) { splineEval(&s->phi, s->phiSpline, r2, &phi, &dphi); e += phi; rb += rho; }
                  else           { dphi = (k->dfE[io] + k->dfE[jo]) * drho; }
                  fx -= dphi * dx; fy -= dphi * dy; fz -= dphi * dz;
               } else if (r2 <= rc2 && r2 > 0.0) {
                  real_t r = sqrt(r2), phi, dphi, rho, drho;
                  tableEval(&s->rho, r, &rho, &drho);
                  if (pass == 1) { tableEval(&s->phi, r, &phi, &dphi); e += phi; rb += rho; }
                  else           { dphi = (k->dfE[io] + k->dfE[jo]) * drho; }
                  fx -= dphi * dx / r; fy -= dphi * dy / r; fz -= dphi * dz / r;
               }
            }
         }
         k->f[0][io] = fx; k->f[1][io] = fy; k->f[2][io] = fz;
         if (pass == 1) { k->U[io] = 0.5 * e; k->rhobar[io] = rb; }
      }
   }
}

/* ---- EAM pass 2: eam.c:352-366, gpu_eam_thread_atom.h:269-287 -------------------------------- */
static void eamEmbed(const OracleSim* s, Rank* k)
{
   real_t ePot = 0.0;
   for (int b = 0; b < k->nLocal; ++b)
      for (int o = b * s->cap, e = o + k->nAtoms[b]; o < e; ++o) {
         real_t F, dF;
         tableEval(&s->F, k->rhobar[o], &F, &dF);
         k->dfE[o] = dF;
         k->U[o] += F;
         ePot += k->U[o];
      }
   k->ePot = ePot;
}

/* ---- haloExchange.c:1811-1860 load/unloadForceBufferCpu, exchangeData for all ranks ------------ */
static void exchangeForce(OracleSim* s)
{
   for (int axis = 0; axis < 3; ++axis) {
      int fm = 2*axis, fp = fm + 1;
      for (int ir = 0; ir < s->nRanks; ++ir) {
         Rank* k = &s->rk[ir];
         for (int side = 0; side < 2; ++side) {
            int face = fm + side, n = 0;
            for (int c = 0; c < k->nForceCells[face]; ++c) {
               int b = k->fSend[face][c];
               for (int o = b * s->cap, e = o + k->nAtoms[b]; o < e; ++o) k->fsend[side][n++] = k->dfE[o];
            }
            k->nfSend[side] = n;
         }
      }
      for (int ir = 0; ir < s->nRanks; ++ir) {
         Rank* k = &s->rk[ir];
         Rank* m = &s->rk[k->nbr[fm]];
         Rank* p = &s->rk[k->nbr[fp]];
         memcpy(m->frecv[1], k->fsend[0], (size_t)k->nfSend[0] * sizeof(real_t)); m->nfRecv[1] = k->nfSend[0];
         memcpy(p->frecv[0], k->fsend[1], (size_t)k->nfSend[1] * sizeof(real_t)); p->nfRecv[0] = k->nfSend[1];
      }
      for (int ir = 0; ir < s->nRanks; ++ir) {
         Rank* k = &s->rk[ir];
         for (int side = 0; side < 2; ++side) {
            int face = fm + side, n = 0;
            for (int c = 0; c < k->nForceCells[face]; ++c) {
               int b = k->fRecv[face][c];
               for (int o = b * s->cap, e = o + k->nAtoms[b]; o < e; ++o) k->dfE[o] = k->frecv[side][n++];
            }
            if (n != k->nfRecv[side]) { fprintf(stderr, "oracle: force exchange count mismatch (%d vs %d)\n", n, k->nfRecv[side]); abort(); }
         }
      }
   }
}

/* ---- timestep.c:102-105 computeForce -> pot->force ------------------------------------------- */
void oracle_use_splines(OracleSim* s)
{
   if (!s->doeam || s->phiSpline) return;
   s->phiSpline = splineCoefficients(&s->phi);
   s->rhoSpline = splineCoefficients(&s->rho);
   oracle_compute_force(s);
   oracle_kinetic_energy(s);                      /* also totals the potential energy of the new forces */
}

void oracle_compute_force(OracleSim* s)
{
   if (!s->doeam) {
      for (int ir = 0; ir < s->nRanks; ++ir) ljForceRank(s, &s->rk[ir]);
      return;
   }
   for (int ir = 0; ir < s->nRanks; ++ir) { eamPass(s, &s->rk[ir], 1); eamEmbed(s, &s->rk[ir]); }
   exchangeForce(s);
   for (int ir = 0; ir < s->nRanks; ++ir) eamPass(s, &s->rk[ir], 3);
}

/* ---- timestep.c:48-100 timestep ----------------------------------------------------------------- */
void oracle_step(OracleSim* s, int nSteps)
{
   double t0 = wallSeconds();
   for (int i = 0; i < nSteps; ++i) {
      oracle_advance_velocity(s, 0.5 * s->dt);
      oracle_advance_position(s, s->dt);
      oracle_redistribute(s);
      oracle_compute_force(s);
      oracle_advance_velocity(s, 0.5 * s->dt);
   }
   oracle_kinetic_energy(s);
   s->loopSeconds += wallSeconds() - t0;
}

/* ---- initAtoms.c:81-124 createFccLattice ---------------------------------------------------------- */
static void createFcc(OracleSim* s, Rank* k)
{
   static const real_t basis[4][3] = { {0.25,0.25,0.25}, {0.25,0.75,0.75}, {0.75,0.25,0.75}, {0.75,0.75,0.25} };
   int lo[3], hi[3];
   for (int a = 0; a < 3; ++a) { lo[a] = (int)floor(k->lmin[a] / s->lat); hi[a] = (int)ceil(k->lmax[a] / s->lat); }
   const real_t zero[3] = { 0.0, 0.0, 0.0 };
   for (int ix = lo[0]; ix < hi[0]; ++ix)
      for (int iy = lo[1]; iy < hi[1]; ++iy)
         for (int iz = lo[2]; iz < hi[2]; ++iz)
            for (int ib = 0; ib < 4; ++ib) {
               real_t r[3] = { (ix + basis[ib][0]) * s->lat, (iy + basis[ib][1]) * s->lat, (iz + basis[ib][2]) * s->lat };
               if (r[0] < k->lmin[0] || r[0] >= k->lmax[0]) continue;
               if (r[1] < k->lmin[1] || r[1] >= k->lmax[1]) continue;
               if (r[2] < k->lmin[2] || r[2] >= k->lmax[2]) continue;
               int gid = ib + 4 * (iz + s->nz * (iy + s->ny * ix));
               putAtom(s, k, gid, 0, r, zero);
            }
}

/* ---- initAtoms.c:162-198 setTemperature (+ setVcm :130-152, computeVcm :220-248) ------------------- */
static void setTemperature(OracleSim* s, real_t temperature)
{
   const real_t mass = s->mass;
   for (int ir = 0; ir < s->nRanks; ++ir) {
      Rank* k = &s->rk[ir];
      for (int b = 0; b < k->nLocal; ++b)
         for (int o = b * s->cap, e = o + k->nAtoms[b]; o < e; ++o) {
            real_t sigma = sqrt(kB_eV * temperature / mass);
            uint64_t seed = oracle_mkSeed((uint32_t)k->gid[o], 123);
            k->p[0][o] = mass * sigma * oracle_gasdev(&seed);
            k->p[1][o] = mass * sigma * oracle_gasdev(&seed);
            k->p[2][o] = mass * sigma * oracle_gasdev(&seed);
         }
   }
   if (temperature == 0.0) return;
   real_t sum[4] = { 0, 0, 0, 0 };
   for (int ir = 0; ir < s->nRanks; ++ir) {
      Rank* k = &s->rk[ir];
      real_t loc[4] = { 0, 0, 0, 0 };
      for (int b = 0; b < k->nLocal; ++b)
         for (int o = b * s->cap, e = o + k->nAtoms[b]; o < e; ++o) {
            loc[0] += k->p[0][o]; loc[1] += k->p[1][o]; loc[2] += k->p[2][o]; loc[3] += mass;
         }
      for (int q = 0; q < 4; ++q) sum[q] += loc[q];
   }
   real_t vShift[3] = { 0.0 - sum[0]/sum[3], 0.0 - sum[1]/sum[3], 0.0 - sum[2]/sum[3] };
   for (int ir = 0; ir < s->nRanks; ++ir) {
      Rank* k = &s->rk[ir];
      for (int b = 0; b < k->nLocal; ++b)
         for (int o = b * s->cap, e = o + k->nAtoms[b]; o < e; ++o)
            for (int a = 0; a < 3; ++a) k->p[a][o] += mass * vShift[a];
   }
   oracle_kinetic_energy(s);
   real_t temp = (s->eKin / s->nGlobal) / kB_eV / 1.5;
   real_t scale = sqrt(temperature / temp);
   for (int ir = 0; ir < s->nRanks; ++ir) {
      Rank* k = &s->rk[ir];
      for (int b = 0; b < k->nLocal; ++b)
         for (int o = b * s->cap, e = o + k->nAtoms[b]; o < e; ++o)
            for (int a = 0; a < 3; ++a) k->p[a][o] *= scale;
   }
   oracle_kinetic_energy(s);
}

/* ---- initAtoms.c:204-216 randomDisplacements -------------------------------------------------------- */
static void randomDisplacements(OracleSim* s, real_t delta)
{
   for (int ir = 0; ir < s->nRanks; ++ir) {
      Rank* k = &s->rk[ir];
      for (int b = 0; b < k->nLocal; ++b)
         for (int o = b * s->cap, e = o + k->nAtoms[b]; o < e; ++o) {
            uint64_t seed = oracle_mkSeed((uint32_t)k->gid[o], 457);
            k->r[0][o] += (2.0 * oracle_lcg61(&seed) - 1.0) * delta;
            k->r[1][o] += (2.0 * oracle_lcg61(&seed) - 1.0) * delta;
            k->r[2][o] += (2.0 * oracle_lcg61(&seed) - 1.0) * delta;
         }
   }
}

static real_t* dalloc(size_t n) { return (real_t*)calloc(n, sizeof(real_t)); }

static void rankInit(OracleSim* s, Rank* k, int ir)
{
   memset(k, 0, sizeof *k);
   int t = ir;
   k->coord[0] = t % s->pg[0]; t /= s->pg[0];
   k->coord[1] = t % s->pg[1];
   k->coord[2] = t / s->pg[1];
   for (int a = 0; a < 3; ++a) {
      k->lext[a] = s->gext[a] / s->pg[a];
      k->lmin[a] = s->gmin[a] +  k->coord[a]      * k->lext[a];
      k->lmax[a] = s->gmin[a] + (k->coord[a] + 1) * k->lext[a];
      k->g[a]    = (int)(k->lext[a] / s->cutoff);
      k->bsize[a] = k->lext[a] / (real_t)k->g[a];
      k->binv[a]  = 1.0 / k->bsize[a];
   }
   k->nLocal = k->g[0] * k->g[1] * k->g[2];
   int nHalo = 2 * ((k->g[0] + 2) * (k->g[1] + k->g[2] + 2) + k->g[1] * k->g[2]);
   k->nTotal = k->nLocal + nHalo;
   k->nAtoms = (int*)calloc((size_t)k->nTotal, sizeof(int));
   size_t slots = (size_t)k->nTotal * s->cap;
   k->gid = (int*)calloc(slots, sizeof(int)); k->spec = (int*)calloc(slots, sizeof(int));
   for (int a = 0; a < 3; ++a) { k->r[a] = dalloc(slots); k->p[a] = dalloc(slots); k->f[a] = dalloc(slots); }
   k->U = dalloc(slots); k->rhobar = dalloc(slots); k->dfE = dalloc(slots);

   /* haloExchange.c:1368-1390 initHaloExchange neighbour ranks; :316-323 PBC shifts */
   k->nbr[0] = rankAt(s, k, -1, 0, 0); k->nbr[1] = rankAt(s, k, +1, 0, 0);
   k->nbr[2] = rankAt(s, k, 0, -1, 0); k->nbr[3] = rankAt(s, k, 0, +1, 0);
   k->nbr[4] = rankAt(s, k, 0, 0, -1); k->nbr[5] = rankAt(s, k, 0, 0, +1);
   for (int a = 0; a < 3; ++a) {
      if (k->coord[a] == 0)             k->shift[2*a][a]   = +1.0 * s->gext[a];
      if (k->coord[a] == s->pg[a] - 1)  k->shift[2*a+1][a] = -1.0 * s->gext[a];
   }
   buildHaloLists(k);
   int maxCells = 0;
   for (int f = 0; f < 6; ++f) if (k->nAtomCells[f] > maxCells) maxCells = k->nAtomCells[f];
   for (int q = 0; q < 2; ++q) {
      k->sendBuf[q] = (AtomRec*)malloc((size_t)maxCells * s->cap * sizeof(AtomRec));
      k->recvBuf[q] = (AtomRec*)malloc((size_t)maxCells * s->cap * sizeof(AtomRec));
      k->fsend[q] = dalloc((size_t)maxCells * s->cap);
      k->frecv[q] = dalloc((size_t)maxCells * s->cap);
   }
   /* neighbour table, self first then getNeighborBoxes order (linkCells.c:202-214) minus self */
   k->nbrBoxes = (int*)malloc((size_t)k->nLocal * 27 * sizeof(int));
   for (int b = 0; b < k->nLocal; ++b) {
      int ix, iy, iz, c = 0; localTuple(k, b, &ix, &iy, &iz);
      k->nbrBoxes[b*27 + c++] = b;
      for (int i = ix-1; i <= ix+1; ++i)
         for (int j = iy-1; j <= iy+1; ++j)
            for (int l = iz-1; l <= iz+1; ++l) {
               int nb = boxFromTuple(k, i, j, l);
               if (nb != b) k->nbrBoxes[b*27 + c++] = nb;
            }
      assert(c == 27);
   }
}

/* LJ cutoff in units of sigma for the simulations created from now on.  The reference hard-wires 5 (ljForce.c:114); upstream CoMD's
 * 2.5 is what its documented LJ cohesive energy (CoMD.c:897) was computed with, so the checker can be run at 2.5 to meet that fixture. */
static double g_ljCutoffSigmas = 5.0;
void oracle_set_lj_cutoff_sigmas(double f) { g_ljCutoffSigmas = f > 0.0 ? f : 5.0; }

OracleSim* oracle_create(int nx, int ny, int nz, int px, int py, int pz,
                         double lat, int doeam, const char* potDir, const char* potName,
                         double temperature, double initialDelta, double dt, int cellCap)
{
   OracleSim* s = (OracleSim*)calloc(1, sizeof *s);
   s->nx = nx; s->ny = ny; s->nz = nz; s->pg[0] = px; s->pg[1] = py; s->pg[2] = pz;
   s->nRanks = px * py * pz; s->doeam = doeam; s->dt = dt;
   if (doeam) {
      /* potType is carried by the file name here: *.alloy = setfl (mycommand.c:286-291 default names) */
      size_t ln = strlen(potName);
      int isSetfl = ln > 6 && strcmp(potName + ln - 6, ".alloy") == 0;
      if ((isSetfl ? readSetfl(s, potDir, potName) : readFuncfl(s, potDir, potName)) != 0) { free(s); return NULL; }
   } else {                                        /* ljForce.c:102-120 */
      s->sigma = 2.315; s->epsilon = 0.167; s->mass = 63.55 * kAmuToInternalMass;
      s->lat = 3.615; s->cutoff = g_ljCutoffSigmas * s->sigma;
   }
   if (lat >= 0.0) s->lat = lat;
   s->cap = cellCap > 0 ? cellCap : (doeam ? 64 : 512);
   const int n[3] = { nx, ny, nz };
   for (int a = 0; a < 3; ++a) {
      s->gmin[a] = 0.0; s->gmax[a] = n[a] * s->lat; s->gext[a] = s->gmax[a] - s->gmin[a];
      /* CoMD.c:571-585 sanityChecks "simulation too small" */
      if (s->gext[a] < 2.0 * s->cutoff * s->pg[a]) {
         fprintf(stderr, "oracle: simulation too small along axis %d\n", a); free(s); return NULL;
      }
   }
   s->nGlobal = 4 * nx * ny * nz;
   s->rk = (Rank*)calloc((size_t)s->nRanks, sizeof(Rank));
   for (int ir = 0; ir < s->nRanks; ++ir) rankInit(s, &s->rk[ir], ir);
   int count = 0;
   for (int ir = 0; ir < s->nRanks; ++ir) {
      createFcc(s, &s->rk[ir]);
      for (int b = 0; b < s->rk[ir].nLocal; ++b) count += s->rk[ir].nAtoms[b];
   }
   if (count != s->nGlobal) { fprintf(stderr, "oracle: lattice holds %d atoms, expected %d\n", count, s->nGlobal); abort(); }
   setTemperature(s, temperature);
   randomDisplacements(s, initialDelta);
   oracle_redistribute(s);
   oracle_compute_force(s);
   oracle_kinetic_energy(s);
   return s;
}

void oracle_destroy(OracleSim* s)
{
   if (s) { free(s->phiSpline); free(s->rhoSpline); }
   if (!s) return;
   for (int ir = 0; ir < s->nRanks; ++ir) {
      Rank* k = &s->rk[ir];
      free(k->nAtoms); free(k->gid); free(k->spec);
      for (int a = 0; a < 3; ++a) { free(k->r[a]); free(k->p[a]); free(k->f[a]); }
      free(k->U); free(k->rhobar); free(k->dfE); free(k->nbrBoxes);
      for (int f = 0; f < 6; ++f) { free(k->atomCells[f]); free(k->fSend[f]); free(k->fRecv[f]); }
      for (int q = 0; q < 2; ++q) { free(k->sendBuf[q]); free(k->recvBuf[q]); free(k->fsend[q]); free(k->frecv[q]); }
   }
   if (s->doeam) { free(s->phi.v - 1); free(s->rho.v - 1); free(s->F.v - 1); }
   free(s->rk); free(s);
}

/* ---- accessors ------------------------------------------------------------------------------------ */
int    oracle_n_global(const OracleSim* s)    { return s->nGlobal; }
int    oracle_n_ranks(const OracleSim* s)     { return s->nRanks; }
double oracle_e_potential(const OracleSim* s) { return s->ePot; }
double oracle_e_kinetic(const OracleSim* s)   { return s->eKin; }
double oracle_cutoff(const OracleSim* s)      { return s->cutoff; }
double oracle_mass(const OracleSim* s)        { return s->mass; }
double oracle_lattice(const OracleSim* s)     { return s->lat; }
double oracle_loop_seconds(const OracleSim* s){ return s->loopSeconds; }
int    oracle_rank_cell_cap(const OracleSim* s) { return s->cap; }

void oracle_set_threads(int n)
{
#ifdef _OPENMP
   if (n > 0) omp_set_num_threads(n);
#else
   (void)n;
#endif
}

int oracle_threads(void)
{
#ifdef _OPENMP
   return omp_get_max_threads();
#else
   return 1;
#endif
}

static real_t* const* vecOf(const Rank* k, int which)
{
   return which == ORACLE_R ? k->r : which == ORACLE_P ? k->p : k->f;
}

void oracle_gather(const OracleSim* s, int which, double* out)
{
   for (int ir = 0; ir < s->nRanks; ++ir) {
      const Rank* k = &s->rk[ir];
      for (int b = 0; b < k->nLocal; ++b)
         for (int o = b * s->cap, e = o + k->nAtoms[b]; o < e; ++o) {
            int g = k->gid[o];
            if (which <= ORACLE_F) { real_t* const* v = vecOf(k, which); for (int a = 0; a < 3; ++a) out[3*g + a] = v[a][o]; }
            else if (which == ORACLE_U)      out[g] = k->U[o];
            else if (which == ORACLE_RHOBAR) out[g] = k->rhobar[o];
            else                             out[g] = k->dfE[o];
         }
   }
}

void oracle_scatter(OracleSim* s, int which, const double* in)
{
   for (int ir = 0; ir < s->nRanks; ++ir) {
      Rank* k = &s->rk[ir];
      real_t* const* v = vecOf(k, which);
      for (int b = 0; b < k->nLocal; ++b)
         for (int o = b * s->cap, e = o + k->nAtoms[b]; o < e; ++o)
            for (int a = 0; a < 3; ++a) v[a][o] = in[3*k->gid[o] + a];
   }
}

void oracle_rank_grid(const OracleSim* s, int rank, int gridSize[3], int* nLocalBoxes, int* nTotalBoxes)
{
   const Rank* k = &s->rk[rank];
   for (int a = 0; a < 3; ++a) gridSize[a] = k->g[a];
   *nLocalBoxes = k->nLocal; *nTotalBoxes = k->nTotal;
}

void oracle_rank_natoms(const OracleSim* s, int rank, int* nAtoms)
{
   memcpy(nAtoms, s->rk[rank].nAtoms, (size_t)s->rk[rank].nTotal * sizeof(int));
}

void oracle_rank_gid(const OracleSim* s, int rank, int* gid)
{
   memcpy(gid, s->rk[rank].gid, (size_t)s->rk[rank].nTotal * s->cap * sizeof(int));
}

void oracle_rank_array(const OracleSim* s, int rank, int which, int comp, double* out)
{
   const Rank* k = &s->rk[rank];
   const real_t* src = which <= ORACLE_F ? vecOf(k, which)[comp]
                     : which == ORACLE_U ? k->U : which == ORACLE_RHOBAR ? k->rhobar : k->dfE;
   const size_t n = (size_t)k->nTotal * s->cap;
   for (size_t i = 0; i < n; ++i) out[i] = src[i];
}

int oracle_box_from_tuple(const OracleSim* s, int rank, int ix, int iy, int iz) { return boxFromTuple(&s->rk[rank], ix, iy, iz); }
int oracle_box_from_coord(const OracleSim* s, int rank, const double r[3])     { const real_t rr[3] = { (real_t)r[0], (real_t)r[1], (real_t)r[2] }; return boxFromCoord(&s->rk[rank], rr); }

int oracle_face_cells(const OracleSim* s, int rank, int kind, int face, int* list)
{
   const Rank* k = &s->rk[rank];
   int n = kind == 0 ? k->nAtomCells[face] : k->nForceCells[face];
   const int* src = kind == 0 ? k->atomCells[face] : kind == 1 ? k->fSend[face] : k->fRecv[face];
   if (list) memcpy(list, src, (size_t)n * sizeof(int));
   return n;
}

int oracle_eam_interpolate(const OracleSim* s, int table, double x, double* f, double* df)
{
   if (!s->doeam) return -1;
   real_t rf, rdf;
   tableEval(table == 0 ? &s->phi : table == 1 ? &s->rho : &s->F, (real_t)x, &rf, &rdf);
   *f = rf; *df = rdf;
   return 0;
}

int oracle_eam_table(const OracleSim* s, int table, int* n, double* x0, double* invDx, double* values)
{
   if (!s->doeam) return -1;
   const Table* t = table == 0 ? &s->phi : table == 1 ? &s->rho : &s->F;
   *n = t->n; *x0 = t->x0; *invDx = t->invDx;
   if (values) for (int i = 0; i < t->n + 3; ++i) values[i] = t->v[i - 1];
   return 0;
}
