/* link_cells.c -- host side of the link-cell grid: sizing (linkCells.c:122-182), cell numbering
 * (comd_geometry.h restates :299-346/:448-480), neighbour stencil (:202-214), occupancy (:412-422).
 * The slot capacity per cell is a run-time value (LinkCell.maxAtoms), not the reference's -DMAXATOMS. */
#include "comd_host.h"
#include <stdlib.h>
#include <assert.h>

/* linkCells.c:28-63 computeHilbertCode: position of (x, y, z) on a 3-D Hilbert curve of order `nbits` (bit-interleaved walk
 * with the per-octant rotations / reflections of the curve) */
static unsigned long hilbertCode(int x, int y, int z, int nbits)
{
   static const int transform[8] = { 0, 1, 7, 6, 3, 2, 4, 5 };
   unsigned long s = 0;
   for (int i = nbits - 1; i >= 0; i--) {
      const int xi = (x >> i) & 1, yi = (y >> i) & 1, zi = (z >> i) & 1;
      const int index = (xi << 2) + (yi << 1) + zi;
      s = (s << 3) + (unsigned long)transform[index];
      int t;
      switch (index) {
         case 0: t = z; z = y; y = t; break;
         case 1: case 5: t = x; x = y; y = t; break;
         case 2: t = ~z; z = ~y; y = t; break;
         case 3: case 7: t = ~x; x = ~y; y = t; break;
         default: x = ~x; z = ~z; break;             /* 4, 6 */
      }
   }
   return s;
}

typedef struct { unsigned long code; int idx; } CodeIdx;
static int cmpCode(const void* a, const void* b)
{
   const CodeIdx* p = (const CodeIdx*)a; const CodeIdx* q = (const CodeIdx*)b;
   return p->code < q->code ? -1 : p->code > q->code ? 1 : (p->idx > q->idx) - (p->idx < q->idx);
}

/* -H (linkCells.c:151-178): number the local cells along a Hilbert curve.  The reference does this for power-of-two grids only
 * (the curve code IS the id there); here the codes of an enclosing power-of-two cube are ranked, so any grid works. */
static void hilbertOrder(LinkCell* ll)
{
   const int gx = ll->gridSize[0], gy = ll->gridSize[1], gz = ll->gridSize[2], n = ll->nLocalBoxes;
   int nbits = 1;
   while ((1 << nbits) < gx || (1 << nbits) < gy || (1 << nbits) < gz) ++nbits;
   CodeIdx* c = (CodeIdx*)malloc((size_t)n * sizeof(CodeIdx));
   for (int iz = 0; iz < gz; ++iz) for (int iy = 0; iy < gy; ++iy) for (int ix = 0; ix < gx; ++ix) {
      const int idx = ix + gx * (iy + gy * iz);
      c[idx].code = hilbertCode(ix, iy, iz, nbits); c[idx].idx = idx;
   }
   qsort(c, (size_t)n, sizeof(CodeIdx), cmpCode);
   ll->boxIDLookUp = (int*)malloc((size_t)n * sizeof(int));
   ll->boxIDLookUpReverse = (int*)malloc((size_t)n * sizeof(int));
   for (int id = 0; id < n; ++id) { ll->boxIDLookUp[c[id].idx] = id; ll->boxIDLookUpReverse[id] = c[id].idx; }
   free(c);
   ll->geom.lookup = ll->boxIDLookUp; ll->geom.reverse = ll->boxIDLookUpReverse;
}

LinkCell* initLinkCells(const Domain* domain, real_t cutoff, int useHilbert)
{
   LinkCell* ll = (LinkCell*)calloc(1, sizeof(LinkCell));
   for (int a = 0; a < 3; ++a) {
      ll->localMin[a] = domain->localMin[a];
      ll->localMax[a] = domain->localMax[a];
      ll->gridSize[a] = (int)(domain->localExtent[a] / cutoff);
      ll->boxSize[a] = domain->localExtent[a] / (real_t)ll->gridSize[a];
      ll->invBoxSize[a] = 1.0 / ll->boxSize[a];
   }
   ll->nLocalBoxes = ll->gridSize[0] * ll->gridSize[1] * ll->gridSize[2];
   ll->nHaloBoxes = 2 * ((ll->gridSize[0] + 2) * (ll->gridSize[1] + ll->gridSize[2] + 2) + ll->gridSize[1] * ll->gridSize[2]);
   ll->nTotalBoxes = ll->nLocalBoxes + ll->nHaloBoxes;
   if (ll->gridSize[0] < 2 || ll->gridSize[1] < 2 || ll->gridSize[2] < 2) {
      fprintf(stderr, "initLinkCells: fewer than 2 link cells along an axis (%d,%d,%d)\n", ll->gridSize[0], ll->gridSize[1], ll->gridSize[2]);
      exit(-1);
   }
   ll->nAtoms = (int*)calloc((size_t)ll->nTotalBoxes, sizeof(int));
   for (int a = 0; a < 3; ++a) {
      ll->geom.g[a] = ll->gridSize[a]; ll->geom.lmin[a] = ll->localMin[a]; ll->geom.lmax[a] = ll->localMax[a]; ll->geom.inv[a] = ll->invBoxSize[a];
   }
   ll->geom.nLocal = ll->nLocalBoxes; ll->geom.nTotal = ll->nTotalBoxes;
   if (useHilbert) hilbertOrder(ll);
   return ll;
}

void destroyLinkCells(LinkCell** boxes)
{
   if (!boxes || !*boxes) return;
   free((*boxes)->nAtoms);
   free((*boxes)->boxIDLookUp); free((*boxes)->boxIDLookUpReverse);
   free(*boxes);
   *boxes = NULL;
}

int getBoxFromTuple(LinkCell* boxes, int x, int y, int z) { return comdBoxFromTuple(&boxes->geom, x, y, z); }
int getBoxFromCoord(LinkCell* boxes, const real_t rr[3]) { return comdBoxFromCoord(&boxes->geom, rr[0], rr[1], rr[2]); }

int getNeighborBoxes(LinkCell* boxes, int iBox, int* nbrBoxes)
{
   int ix, iy, iz, count = 0;
   comdTupleFromBox(&boxes->geom, iBox, &ix, &iy, &iz);
   for (int i = ix - 1; i <= ix + 1; i++)
      for (int j = iy - 1; j <= iy + 1; j++)
         for (int k = iz - 1; k <= iz + 1; k++)
            nbrBoxes[count++] = getBoxFromTuple(boxes, i, j, k);
   return count;
}

int maxOccupancy(LinkCell* boxes)
{
   int localMax = 0, globalMax;
   for (int i = 0; i < boxes->nLocalBoxes; ++i) if (boxes->nAtoms[i] > localMax) localMax = boxes->nAtoms[i];
   startTimer(commReduceTimer);
   maxIntParallel(&localMax, &globalMax, 1);
   stopTimer(commReduceTimer);
   return globalMax;
}
