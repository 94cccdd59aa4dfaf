/* domain.c -- 3-D Cartesian spatial decomposition: one equal brick per rank, periodic neighbours.
 * Behaviour of decomposition.c:18-66 (rank = ix + px*(iy + py*iz)). */
#include "comd_host.h"
#include <stdlib.h>

Domain* initDecomposition(int xproc, int yproc, int zproc, const real_t globalExtent[3])
{
   if (xproc * yproc * zproc != getNRanks()) {
      fprintf(stderr, "initDecomposition: grid %dx%dx%d does not match %d ranks\n", xproc, yproc, zproc, getNRanks());
      exit(-1);
   }
   Domain* dd = (Domain*)calloc(1, sizeof(Domain));
   dd->procGrid[0] = xproc; dd->procGrid[1] = yproc; dd->procGrid[2] = zproc;
   int r = getMyRank();
   dd->procCoord[0] = r % xproc; r /= xproc;
   dd->procCoord[1] = r % yproc;
   dd->procCoord[2] = r / yproc;
   for (int a = 0; a < 3; ++a) {
      dd->globalMin[a] = 0;
      dd->globalMax[a] = globalExtent[a];
      dd->globalExtent[a] = dd->globalMax[a] - dd->globalMin[a];
      dd->localExtent[a] = dd->globalExtent[a] / dd->procGrid[a];
      dd->localMin[a] = dd->globalMin[a] +  dd->procCoord[a]      * dd->localExtent[a];
      dd->localMax[a] = dd->globalMin[a] + (dd->procCoord[a] + 1) * dd->localExtent[a];
   }
   return dd;
}

int processorNum(Domain* domain, int dix, int diy, int diz)
{
   const int* c = domain->procCoord; const int* g = domain->procGrid;
   int ix = (c[0] + dix + g[0]) % g[0];
   int iy = (c[1] + diy + g[1]) % g[1];
   int iz = (c[2] + diz + g[2]) % g[2];
   return ix + g[0] * (iy + g[1] * iz);
}
