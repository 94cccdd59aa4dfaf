/* comd_geometry.h -- link-cell index arithmetic shared by the C host and the HIP kernels.
 *
 * Restates the reference's halo-cell numbering and position->cell rules so that host, device and
 * oracle agree on every cell id: linkCells.c:299-346 (getBoxFromTuple), :448-480 (getBoxFromCoord,
 * including its tie-breaking at the upper domain face) and their device copies in
 * gpu_redistribute.h:39-133.  Plain C99 / HIP; no dependencies.
 */
#ifndef COMD_GEOMETRY_H
#define COMD_GEOMETRY_H

#include <math.h>

#if defined(__HIPCC__)
#define COMD_HD __host__ __device__ static inline
#else
#define COMD_HD static inline
#endif

/* positions and cell geometry carry the build's precision (mytype.h:8-21): the cell an atom lands in is decided by real_t arithmetic,
 * identically on the host, on the device and in the checker */
#ifdef COMD_SINGLE
typedef float comd_real;
#else
typedef double comd_real;
#endif

typedef struct CellGeom {
   int    g[3];            /* local grid */
   int    nLocal, nTotal;
   comd_real lmin[3], lmax[3], inv[3];
   /* optional renumbering of the LOCAL cells (space-filling curve, linkCells.c:160-178 boxIDLookUp / boxIDLookUpReverse):
    * lookup[ix + gx*(iy + gy*iz)] = cell id, reverse[id] = ix + gx*(iy + gy*iz).  NULL: natural order.  Halo cells keep their numbers. */
   const int* lookup;
   const int* reverse;
} CellGeom;

/* Cell id of grid tuple (ix,iy,iz), each in [-1, g]; halo slabs are numbered after the local cells
 * in the order x-, x+, y-, y+, z-, z+ with the y slabs spanning the x halos and the z slabs both. */
COMD_HD int comdBoxFromTuple(const CellGeom* c, int ix, int iy, int iz)
{
   const int gx = c->g[0], gy = c->g[1], gz = c->g[2];
   const int base = c->nLocal;
   if (iz == gz) return base + 2*gz*gy + 2*gz*(gx+2) + (gx+2)*(gy+2) + (gx+2)*(iy+1) + (ix+1);
   if (iz == -1) return base + 2*gz*gy + 2*gz*(gx+2) + (gx+2)*(iy+1) + (ix+1);
   if (iy == gy) return base + 2*gz*gy + gz*(gx+2) + (gx+2)*iz + (ix+1);
   if (iy == -1) return base + 2*gz*gy + iz*(gx+2) + (ix+1);
   if (ix == gx) return base + gy*gz + iz*gy + iy;
   if (ix == -1) return base + iz*gy + iy;
   const int idx = ix + gx*(iy + gy*iz);
   return c->lookup ? c->lookup[idx] : idx;
}

/* Cell that owns position (x,y,z).  Inside the local domain the result is always a local cell (an atom
 * that rounds onto the upper face stays in the last cell); at or beyond localMax it is the halo cell. */
COMD_HD int comdBoxFromCoord(const CellGeom* c, comd_real x, comd_real y, comd_real z)
{
   int ix = (int)floor((double)((x - c->lmin[0]) * c->inv[0]));
   int iy = (int)floor((double)((y - c->lmin[1]) * c->inv[1]));
   int iz = (int)floor((double)((z - c->lmin[2]) * c->inv[2]));
   if (x < c->lmax[0]) { if (ix == c->g[0]) ix = c->g[0] - 1; } else ix = c->g[0];
   if (y < c->lmax[1]) { if (iy == c->g[1]) iy = c->g[1] - 1; } else iy = c->g[1];
   if (z < c->lmax[2]) { if (iz == c->g[2]) iz = c->g[2] - 1; } else iz = c->g[2];
   return comdBoxFromTuple(c, ix, iy, iz);
}

/* 1 when every tuple component lies in [-1, g]: an atom further out than the halo has been lost. */
COMD_HD int comdCoordInHalo(const CellGeom* c, comd_real x, comd_real y, comd_real z)
{
   int ix = (int)floor((double)((x - c->lmin[0]) * c->inv[0]));
   int iy = (int)floor((double)((y - c->lmin[1]) * c->inv[1]));
   int iz = (int)floor((double)((z - c->lmin[2]) * c->inv[2]));
   return ix >= -1 && iy >= -1 && iz >= -1 && ix <= c->g[0] && iy <= c->g[1] && iz <= c->g[2];
}

/* inverse of comdBoxFromTuple (linkCells.c:497-568 getTuple) */
COMD_HD void comdTupleFromBox(const CellGeom* c, int iBox, int* ixp, int* iyp, int* izp)
{
   const int gx = c->g[0], gy = c->g[1], gz = c->g[2];
   int ix, iy, iz;
   if (iBox < c->nLocal) {
      const int idx = c->reverse ? c->reverse[iBox] : iBox;
      ix = idx % gx; iy = (idx / gx) % gy; iz = idx / (gx*gy);
   } else {
      int ink = iBox - c->nLocal;
      if (ink < 2*gy*gz) {
         if (ink < gy*gz) ix = -1; else { ink -= gy*gz; ix = gx; }
         iy = ink % gy; iz = ink / gy;
      } else if (ink < 2*gz*(gy + gx + 2)) {
         ink -= 2*gz*gy;
         if (ink < (gx+2)*gz) iy = -1; else { ink -= (gx+2)*gz; iy = gy; }
         ix = ink % (gx+2) - 1; iz = ink / (gx+2);
      } else {
         ink -= 2*gz*(gy + gx + 2);
         if (ink < (gx+2)*(gy+2)) iz = -1; else { ink -= (gx+2)*(gy+2); iz = gz; }
         ix = ink % (gx+2) - 1; iy = ink / (gx+2) - 1;
      }
   }
   *ixp = ix; *iyp = iy; *izp = iz;
}

#endif
