/* link_cells.c -- host side of the link-cell grid: sizing (linkCells.c:122-182), cell numbering
 * (comd_geometry.h restates :299-346/:448-480), neighbour stencil (:202-214), occupancy (:412-422).
 * The slot capacity per cell is a run-time value (LinkCell.maxAtoms), not the reference's -DMAXATOMS. */
#include "comd_host.h"
#include <stdlib.h>
#include <assert.h>

LinkCell* initLinkCells(const Domain* domain, real_t cutoff)
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
   return ll;
}

void destroyLinkCells(LinkCell** boxes)
{
   if (!boxes || !*boxes) return;
   free((*boxes)->nAtoms);
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
