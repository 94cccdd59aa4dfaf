/* halo_exchange.c -- six-face halo exchange, three sequential axis phases (x, y, z), two messages per phase.
 *
 * Same protocol as the reference (haloExchange.c:8-29, exchangeData :1493-1522): per axis pack the minus and plus
 * faces, exchange both, then unpack minus and plus -- the order that keeps migrating atoms from being duplicated.
 * Same cell lists (mkAtomCellList :1543-1567, mkForceSend/RecvCellList :1712-1801), same periodic shifts (:316-323),
 * same HaloExchange {loadBuffer, unloadBuffer, destroy} plugin shape (haloExchange.h:84-104).
 *
 * What differs: messages never leave the device.  loadBuffer runs the pack kernels into a device buffer, the
 * transport moves device memory (RCCL send/recv over xGMI between ranks; nothing at all when a rank is its own
 * periodic neighbour), unloadBuffer runs the unpack kernels.  The reference stages every message through pinned
 * host memory and blocks on six count read-backs per exchange (haloExchange.c:1632-1633, gpu_kernels.cu:534-535).
 * The whole GPUDirect-Async / libmp machinery of haloExchange.c:498-1366 has no counterpart by design.
 *
 * Message sizes are data dependent and RCCL cannot probe.  Instead of a size handshake (and the host synchronisations it needs) per
 * axis phase, both ends of a message derive its transfer size from the count of the SAME message one step earlier, which both hold
 * (the sender packed it, the receiver unpacked it; a one-wave kernel mirrors the four counts of a phase into pinned memory):
 * count + 12.5 % + 64 atoms, the true count travels in the message.  The first exchange, the exchanges after a Verlet-list build and
 * runs with COMD_HALO_HANDSHAKE=1 swap exact sizes instead.  A message that outgrows its agreed size raises the device status flag
 * (pack and unpack kernels both check) and the run stops at the next status read.
 */
#include "comd_host.h"
#include <stdlib.h>
#include <string.h>

#define MAXI(a, b) ((a) > (b) ? (a) : (b))

static HaloExchange* initHaloExchangeBase(Domain* domain)
{
   HaloExchange* hh = (HaloExchange*)calloc(1, sizeof(HaloExchange));
   hh->nbrRank[HALO_X_MINUS] = processorNum(domain, -1,  0,  0);
   hh->nbrRank[HALO_X_PLUS]  = processorNum(domain, +1,  0,  0);
   hh->nbrRank[HALO_Y_MINUS] = processorNum(domain,  0, -1,  0);
   hh->nbrRank[HALO_Y_PLUS]  = processorNum(domain,  0, +1,  0);
   hh->nbrRank[HALO_Z_MINUS] = processorNum(domain,  0,  0, -1);
   hh->nbrRank[HALO_Z_PLUS]  = processorNum(domain,  0,  0, +1);
   return hh;
}

static int* cellBlock(LinkCell* boxes, const int lo[3], const int hi[3], int nCells)
{
   int* list = (int*)malloc((size_t)nCells * sizeof(int));
   int count = 0;
   for (int ix = lo[0]; ix < hi[0]; ++ix)
      for (int iy = lo[1]; iy < hi[1]; ++iy)
         for (int iz = lo[2]; iz < hi[2]; ++iz)
            list[count++] = getBoxFromTuple(boxes, ix, iy, iz);
   if (count != nCells) { fprintf(stderr, "halo cell list: built %d cells, expected %d\n", count, nCells); exit(-1); }
   return list;
}

/* atoms: the halo plane and the first local plane of the face, over the full (halo-inclusive) extent of the other
 * two axes, so that data received on earlier axes is forwarded to edge and corner neighbours */
int* mkAtomCellList(LinkCell* boxes, enum HaloFaceOrder iFace, int nCells)
{
   int lo[3] = { -1, -1, -1 }, hi[3] = { boxes->gridSize[0] + 1, boxes->gridSize[1] + 1, boxes->gridSize[2] + 1 };
   const int axis = iFace / 2;
   if (iFace & 1) lo[axis] = hi[axis] - 2; else hi[axis] = lo[axis] + 2;
   return cellBlock(boxes, lo, hi, nCells);
}

static void forceBlock(LinkCell* boxes, int face, int recv, int lo[3], int hi[3])
{
   const int axis = face / 2;
   for (int a = 0; a < 3; ++a) {
      if (a < axis) { lo[a] = -1; hi[a] = boxes->gridSize[a] + 1; }    /* axes already exchanged: their halos ride along */
      else          { lo[a] = 0;  hi[a] = boxes->gridSize[a]; }
   }
   const int g = boxes->gridSize[axis];
   if (!recv) { if (face & 1) { lo[axis] = g - 1; hi[axis] = g; } else { lo[axis] = 0; hi[axis] = 1; } }
   else       { if (face & 1) { lo[axis] = g; hi[axis] = g + 1; } else { lo[axis] = -1; hi[axis] = 0; } }
}

int* mkForceSendCellList(LinkCell* boxes, int face, int nCells)
{
   int lo[3], hi[3]; forceBlock(boxes, face, 0, lo, hi);
   return cellBlock(boxes, lo, hi, nCells);
}

int* mkForceRecvCellList(LinkCell* boxes, int face, int nCells)
{
   int lo[3], hi[3]; forceBlock(boxes, face, 1, lo, hi);
   return cellBlock(boxes, lo, hi, nCells);
}

static int* uploadInts(const int* h, int n)
{
   int* d = (int*)comdDeviceMalloc((long)n * sizeof(int));
   comdMemcpyHtoD(d, h, (long)n * sizeof(int));
   return d;
}

/* ---- atom exchange plugin ---------------------------------------------------------------------------- */
static int loadAtomsBuffer(void* vparms, void* data, int face, char* buf)
{
   AtomExchangeParms* parms = (AtomExchangeParms*)vparms;
   SimFlat* sim = (SimFlat*)data;
   const int bound = parms->sendBound[face] > 0 && parms->sendBound[face] < parms->capacityAtoms ? parms->sendBound[face] : parms->capacityAtoms;
   compactCellsGpu(buf, parms->nCells[face], parms->cellListGpu[face], &sim->gpu, parms->d_cellOffsets,
                   parms->shift[face], bound, sim->gpu.boundary_stream);
   return -1;            /* the count stays on the device (message header); exchangeData asks msgBytes only if a peer needs it */
}

/* blocking: the reference reads this count back after every pack (gpu_kernels.cu:534-535); here only when the message leaves the rank,
 * and after BOTH faces of the axis have been packed, so one stream drain serves the two reads */
static int atomsMsgBytes(void* vparms, void* data, int face, char* buf)
{
   (void)vparms; (void)face;
   SimFlat* sim = (SimFlat*)data;
   int n = atomMsgCountGpu(&sim->gpu, buf, sim->gpu.boundary_stream);
   return COMD_ATOM_MSG_HEADER + n * COMD_ATOM_MSG_BYTES_PER_ATOM;
}

static void atomsMsgBytes2(void* vparms, void* data, int faceM, char* bufM, int faceP, char* bufP, int out[2])
{
   (void)vparms; (void)faceM; (void)faceP;
   SimFlat* sim = (SimFlat*)data;
   int n[2];
   comdReadDeviceInt2((const int*)bufM, (const int*)bufP, n, sim->gpu.boundary_stream);      /* the counts sit in the message headers */
   out[0] = COMD_ATOM_MSG_HEADER + n[0] * COMD_ATOM_MSG_BYTES_PER_ATOM;
   out[1] = COMD_ATOM_MSG_HEADER + n[1] * COMD_ATOM_MSG_BYTES_PER_ATOM;
}

static void unloadAtomsBuffer(void* vparms, void* data, int face, int bufSize, char* buf)
{
   AtomExchangeParms* parms = (AtomExchangeParms*)vparms;
   SimFlat* sim = (SimFlat*)data;
   int nBuf = bufSize < 0 ? -1 : (bufSize - COMD_ATOM_MSG_HEADER) / COMD_ATOM_MSG_BYTES_PER_ATOM;
   const int bound = nBuf < 0 && parms->recvBound[face] > 0 && parms->recvBound[face] < parms->capacityAtoms ? parms->recvBound[face] : parms->capacityAtoms;
   unloadAtomsBufferToGpu(buf, nBuf, bound, &sim->gpu, sim->gpu.boundary_stream);
}

static void loadAtomsBuffer2(void* vparms, void* data, int faceM, char* bufM, int faceP, char* bufP, int nBytes[2])
{
   AtomExchangeParms* parms = (AtomExchangeParms*)vparms;
   SimFlat* sim = (SimFlat*)data;
   const int f[2] = { faceM, faceP };
   int nCells[2], caps[2]; int* lists[2]; int* offs[2] = { parms->d_cellOffsets, parms->d_cellOffsets2 };
   for (int k = 0; k < 2; ++k) {
      nCells[k] = parms->nCells[f[k]]; lists[k] = parms->cellListGpu[f[k]];
      caps[k] = parms->sendBound[f[k]] > 0 && parms->sendBound[f[k]] < parms->capacityAtoms ? parms->sendBound[f[k]] : parms->capacityAtoms;
   }
   compactCellsGpu2(bufM, bufP, nCells, lists, &sim->gpu, offs, parms->shift[faceM], parms->shift[faceP], caps, sim->gpu.boundary_stream);
   nBytes[0] = nBytes[1] = -1;              /* counts stay on the device (message headers) */
}

static void unloadAtomsBuffer2(void* vparms, void* data, int faceA, int bufSizeA, char* bufA, int faceB, int bufSizeB, char* bufB)
{
   AtomExchangeParms* parms = (AtomExchangeParms*)vparms;
   SimFlat* sim = (SimFlat*)data;
   const int f[2] = { faceA, faceB }, size[2] = { bufSizeA, bufSizeB };
   int nBuf[2], bound[2];
   for (int k = 0; k < 2; ++k) {
      nBuf[k] = size[k] < 0 ? -1 : (size[k] - COMD_ATOM_MSG_HEADER) / COMD_ATOM_MSG_BYTES_PER_ATOM;
      bound[k] = nBuf[k] < 0 && parms->recvBound[f[k]] > 0 && parms->recvBound[f[k]] < parms->capacityAtoms ? parms->recvBound[f[k]] : parms->capacityAtoms;
   }
   unloadAtomsBufferToGpu2(bufA, nBuf[0], bound[0], bufB, nBuf[1], bound[1], &sim->gpu, sim->gpu.boundary_stream);
}

static void atomsCountPtrs(void* vparms, HaloExchange* hh, int faceM, int faceP, const int* out[4])
{
   (void)vparms; (void)faceM; (void)faceP;         /* the counts sit in the message headers */
   out[0] = (const int*)hh->sendBufM; out[1] = (const int*)hh->sendBufP; out[2] = (const int*)hh->recvBufP; out[3] = (const int*)hh->recvBufM;
}

static void atomsSetBounds(void* vparms, int face, int sendBoundAtoms, int recvBoundAtoms)
{
   AtomExchangeParms* parms = (AtomExchangeParms*)vparms;
   parms->sendBound[face] = sendBoundAtoms; parms->recvBound[face] = recvBoundAtoms;
}

static void destroyAtomsExchange(void* vparms)
{
   AtomExchangeParms* parms = (AtomExchangeParms*)vparms;
   for (int f = 0; f < 6; ++f) { free(parms->cellList[f]); comdDeviceFree(parms->cellListGpu[f]); }
   comdDeviceFree(parms->d_cellOffsets); comdDeviceFree(parms->d_cellOffsets2);
}

/* self-neighbour tail of the atom exchange: one launch per axis, in order (an axis re-sends what the earlier ones brought in) */
static void atomsMirror(HaloExchange* hh, void* vdata, int firstAxis)
{
   AtomExchangeParms* parms = (AtomExchangeParms*)hh->parms;
   SimFlat* sim = (SimFlat*)vdata;
   for (int a = firstAxis; a < 3; ++a) {
      const int n[2] = { parms->nCells[2 * a], parms->nCells[2 * a + 1] };
      int* const lists[2] = { parms->cellListGpu[2 * a], parms->cellListGpu[2 * a + 1] };
      mirrorAtomCellsGpu(n, lists, parms->shift[2 * a], parms->shift[2 * a + 1], firstAxis, a, &sim->gpu, sim->gpu.boundary_stream);
   }
}

HaloExchange* initAtomHaloExchange(Domain* domain, LinkCell* boxes, int allocDevice)
{
   HaloExchange* hh = initHaloExchangeBase(domain);
   const int* g = boxes->gridSize;
   const int size0 = (g[1] + 2) * (g[2] + 2), size1 = (g[0] + 2) * (g[2] + 2), size2 = (g[0] + 2) * (g[1] + 2);
   const int maxSize = MAXI(size0, MAXI(size1, size2));       /* the reference forgets size0 (haloExchange.c:205-206) */
   AtomExchangeParms* parms = (AtomExchangeParms*)calloc(1, sizeof(AtomExchangeParms));
   parms->capacityAtoms = maxSize * 2 * boxes->maxAtoms;
   hh->bufCapacity = COMD_ATOM_MSG_HEADER + parms->capacityAtoms * COMD_ATOM_MSG_BYTES_PER_ATOM;
   hh->loadBuffer = loadAtomsBuffer;
   hh->msgBytes = atomsMsgBytes;
   hh->msgBytes2 = atomsMsgBytes2;
   hh->unloadBuffer = unloadAtomsBuffer;
   hh->destroy = destroyAtomsExchange;
   parms->nCells[HALO_X_MINUS] = parms->nCells[HALO_X_PLUS] = 2 * size0;
   parms->nCells[HALO_Y_MINUS] = parms->nCells[HALO_Y_PLUS] = 2 * size1;
   parms->nCells[HALO_Z_MINUS] = parms->nCells[HALO_Z_PLUS] = 2 * size2;
   for (int f = 0; f < 6; ++f) parms->cellList[f] = mkAtomCellList(boxes, (enum HaloFaceOrder)f, parms->nCells[f]);
   for (int a = 0; a < 3; ++a) {
      if (domain->procCoord[a] == 0)                       parms->shift[2*a][a]     = +1.0 * domain->globalExtent[a];
      if (domain->procCoord[a] == domain->procGrid[a] - 1) parms->shift[2*a + 1][a] = -1.0 * domain->globalExtent[a];
   }
   hh->type = 0;
   hh->parms = parms;
   hh->deviceBuffers = allocDevice;
   hh->nTotalBoxes = boxes->nTotalBoxes; if (allocDevice) hh->mirror = atomsMirror;
   hh->msgHeaderBytes = COMD_ATOM_MSG_HEADER; hh->msgBytesPerAtom = COMD_ATOM_MSG_BYTES_PER_ATOM; hh->capacityAtoms = parms->capacityAtoms;
   hh->countPtrs = atomsCountPtrs; hh->setBounds = atomsSetBounds;
   if (allocDevice) {
      for (int f = 0; f < 6; ++f) parms->cellListGpu[f] = uploadInts(parms->cellList[f], parms->nCells[f]);
      parms->d_cellOffsets = (int*)comdDeviceMalloc((long)(2 * maxSize + 1) * sizeof(int));
      parms->d_cellOffsets2 = (int*)comdDeviceMalloc((long)(2 * maxSize + 1) * sizeof(int));
      hh->loadBuffer2 = loadAtomsBuffer2; hh->unloadBuffer2 = unloadAtomsBuffer2;
      hh->sendBufM = (char*)comdDeviceMalloc(hh->bufCapacity); hh->sendBufP = (char*)comdDeviceMalloc(hh->bufCapacity);
      hh->recvBufM = (char*)comdDeviceMalloc(hh->bufCapacity); hh->recvBufP = (char*)comdDeviceMalloc(hh->bufCapacity);
   }
   return hh;
}

/* ---- force (dfEmbed) exchange plugin --------------------------------------------------------------------- */
static int loadForceBuffer(void* vparms, void* vdata, int face, char* buf)
{
   ForceExchangeParms* parms = (ForceExchangeParms*)vparms;
   SimFlat* s = (SimFlat*)vdata;
   s->gpu.msgBoundAtoms = parms->sendBound[face];
   loadForceBufferFromGpu((real_t*)buf, parms->nCells[face], parms->sendCellsGpu[face], parms->sendOffsetsGpu[face], &s->gpu, s->gpu.boundary_stream);
   s->gpu.msgBoundAtoms = 0;
   return -1;
}

static void loadSlotBuffer2(void* vparms, void* vdata, int faceM, char* bufM, int faceP, char* bufP, int nBytes[2])
{
   ForceExchangeParms* parms = (ForceExchangeParms*)vparms;
   SimFlat* s = (SimFlat*)vdata;
   const int f[2] = { faceM, faceP };
   int nCells[2], bounds[2]; int* lists[2]; int* offs[2];
   for (int k = 0; k < 2; ++k) { nCells[k] = parms->nCells[f[k]]; lists[k] = parms->sendCellsGpu[f[k]]; offs[k] = parms->sendOffsetsGpu[f[k]]; bounds[k] = parms->sendBound[f[k]]; }
   if (parms->positions) {
      loadPositionBufferFromGpu2((real_t*)bufM, (real_t*)bufP, nCells, lists, offs, bounds, parms->shift[faceM], parms->shift[faceP], &s->gpu, s->gpu.boundary_stream);
      nBytes[0] = parms->msgBytesCached[faceM]; nBytes[1] = parms->msgBytesCached[faceP];
   } else {
      loadForceBufferFromGpu2((real_t*)bufM, (real_t*)bufP, nCells, lists, offs, bounds, &s->gpu, s->gpu.boundary_stream);
      nBytes[0] = nBytes[1] = -1;
   }
}

static void unloadSlotBuffer2(void* vparms, void* vdata, int faceA, int bufSizeA, char* bufA, int faceB, int bufSizeB, char* bufB)
{
   ForceExchangeParms* parms = (ForceExchangeParms*)vparms;
   SimFlat* s = (SimFlat*)vdata;
   (void)bufSizeA; (void)bufSizeB;         /* positional */
   const int f[2] = { faceA, faceB };
   int nCells[2]; int* lists[2]; int* offs[2];
   for (int k = 0; k < 2; ++k) { nCells[k] = parms->nCells[f[k]]; lists[k] = parms->recvCellsGpu[f[k]]; offs[k] = parms->recvOffsetsGpu[f[k]]; }
   if (parms->positions) unloadPositionBufferToGpu2((const real_t*)bufA, (const real_t*)bufB, nCells, lists, offs, &s->gpu, s->gpu.boundary_stream);
   else                  unloadForceBufferToGpu2((const real_t*)bufA, (const real_t*)bufB, nCells, lists, offs, &s->gpu, s->gpu.boundary_stream);
}

/* force and position messages: the counts are the totals of the batched scan (send lists: what leaves; receive lists: what must arrive) */
static void forceCountPtrs(void* vparms, HaloExchange* hh, int faceM, int faceP, const int* out[4])
{
   (void)hh;
   ForceExchangeParms* parms = (ForceExchangeParms*)vparms;
   out[0] = parms->sendOffsetsGpu[faceM] + parms->nCells[faceM]; out[1] = parms->sendOffsetsGpu[faceP] + parms->nCells[faceP];
   out[2] = parms->recvOffsetsGpu[faceP] + parms->nCells[faceP]; out[3] = parms->recvOffsetsGpu[faceM] + parms->nCells[faceM];
}

static void forceSetBounds(void* vparms, int face, int sendBoundAtoms, int recvBoundAtoms)
{
   (void)recvBoundAtoms;                           /* positional unpack: the receive cells know their own occupancy */
   ((ForceExchangeParms*)vparms)->sendBound[face] = sendBoundAtoms;
}

static int forceMsgBytes(void* vparms, void* vdata, int face, char* buf)
{
   (void)buf;
   ForceExchangeParms* parms = (ForceExchangeParms*)vparms;
   SimFlat* s = (SimFlat*)vdata;
   int n = comdReadDeviceInt(parms->sendOffsetsGpu[face] + parms->nCells[face], s->gpu.boundary_stream);
   return n * (int)sizeof(real_t);
}

static void forceMsgBytes2(void* vparms, void* vdata, int faceM, char* bufM, int faceP, char* bufP, int out[2])
{
   (void)bufM; (void)bufP;
   ForceExchangeParms* parms = (ForceExchangeParms*)vparms;
   SimFlat* s = (SimFlat*)vdata;
   int n[2];
   comdReadDeviceInt2(parms->sendOffsetsGpu[faceM] + parms->nCells[faceM], parms->sendOffsetsGpu[faceP] + parms->nCells[faceP], n, s->gpu.boundary_stream);
   out[0] = n[0] * (int)sizeof(real_t); out[1] = n[1] * (int)sizeof(real_t);
}

static void unloadForceBuffer(void* vparms, void* vdata, int face, int bufSize, char* buf)
{
   ForceExchangeParms* parms = (ForceExchangeParms*)vparms;
   SimFlat* s = (SimFlat*)vdata;
   (void)bufSize;       /* positional: the receive cells hold the same atoms, in gid order, as the sender's send cells */
   unloadForceBufferToGpu((const real_t*)buf, parms->nCells[face], parms->recvCellsGpu[face], parms->recvOffsetsGpu[face], &s->gpu, s->gpu.boundary_stream);
}

static void destroyForceExchange(void* vparms)
{
   ForceExchangeParms* parms = (ForceExchangeParms*)vparms;
   for (int f = 0; f < 6; ++f) {
      free(parms->sendCells[f]); free(parms->recvCells[f]);
      comdDeviceFree(parms->sendCellsGpu[f]); comdDeviceFree(parms->recvCellsGpu[f]);
      comdDeviceFree(parms->sendOffsetsGpu[f]); comdDeviceFree(parms->recvOffsetsGpu[f]);
   }
   comdDeviceFree(parms->d_cellOffsets);
}

/* [round 4] Self-neighbour axes at the END of the x -> y -> z sequence (all three on one rank; y and z on a 2 x 1 x 1 grid; z on 2 x 2 x 1): every halo
 * cell they fill is the periodic image of one known cell.  What a rank sends through face F arrives at its own opposite face (haloExchange.c:788-853), and a
 * later axis re-sends the halo cells of the earlier ones (:1504-1520), so an edge or corner cell is the image of an image: the chain is folded here, once,
 * into (halo cell, source cell, total shift) triples -- a coordinate is shifted by at most one axis of the chain, so adding the total is the same arithmetic --
 * and ONE launch (comd_hip.h mirrorSlotCellsGpu) replaces a pack and an unpack per axis.  COMD_HALO_MIRROR=0 keeps the message path (A/B runs). */
int haloMirrorFirstAxis(const HaloExchange* hh)
{
   static int enabled = -1;
   if (enabled < 0) { const char* e = getenv("COMD_HALO_MIRROR"); enabled = !(e && atoi(e) == 0); }
   if (!enabled || !hh->mirror || !hh->deviceBuffers || loopbackParallel()) return 3;
   int first = 3;
   for (int a = 2; a >= 0; --a) {
      if (hh->nbrRank[2 * a] == getMyRank() && hh->nbrRank[2 * a + 1] == getMyRank()) first = a; else break;
   }
   return first;
}

static void slotMirror(HaloExchange* hh, void* vdata, int firstAxis)
{
   ForceExchangeParms* parms = (ForceExchangeParms*)hh->parms;
   SimFlat* s = (SimFlat*)vdata;
   if (!parms->mirrorDstGpu || parms->mirrorFirst != firstAxis) {
      int n = 0;
      for (int a = firstAxis; a < 3; ++a) n += 2 * parms->nCells[2 * a];
      int* dst = (int*)malloc((size_t)n * sizeof(int)); int* src = (int*)malloc((size_t)n * sizeof(int));
      real_t* sh = (real_t*)calloc((size_t)3 * n, sizeof(real_t));
      int* rootOf = (int*)malloc((size_t)hh->nTotalBoxes * sizeof(int));
      real_t* shiftOf = (real_t*)calloc((size_t)3 * hh->nTotalBoxes, sizeof(real_t));
      for (int c = 0; c < hh->nTotalBoxes; ++c) rootOf[c] = -1;
      int k = 0;
      for (int a = firstAxis; a < 3; ++a)
         for (int dir = 0; dir < 2; ++dir) {
            const int faceSend = 2 * a + dir, faceRecv = 2 * a + (1 - dir);     /* what leaves through a face arrives at the opposite one */
            for (int i = 0; i < parms->nCells[faceSend]; ++i, ++k) {
               int from = parms->sendCells[faceSend][i];
               const int to = parms->recvCells[faceRecv][i];
               real_t t[3] = { parms->shift[faceSend][0], parms->shift[faceSend][1], parms->shift[faceSend][2] };
               if (rootOf[from] >= 0) { for (int d = 0; d < 3; ++d) t[d] += shiftOf[3 * from + d]; from = rootOf[from]; }      /* a halo cell of an earlier axis of the chain */
               rootOf[to] = from; for (int d = 0; d < 3; ++d) shiftOf[3 * to + d] = t[d];
               dst[k] = to; src[k] = from; for (int d = 0; d < 3; ++d) sh[3 * k + d] = t[d];
            }
         }
      if (parms->mirrorDstGpu) { comdDeviceFree(parms->mirrorDstGpu); comdDeviceFree(parms->mirrorSrcGpu); comdDeviceFree(parms->mirrorShiftGpu); }
      parms->mirrorDstGpu = uploadInts(dst, n); parms->mirrorSrcGpu = uploadInts(src, n);
      parms->mirrorShiftGpu = (real_t*)comdDeviceMalloc((long)3 * n * sizeof(real_t));
      comdMemcpyHtoD(parms->mirrorShiftGpu, sh, (long)3 * n * sizeof(real_t));
      parms->mirrorFirst = firstAxis; parms->mirrorPairs = n;
      free(dst); free(src); free(sh); free(rootOf); free(shiftOf);
   }
   mirrorSlotCellsGpu(parms->positions ? 1 : 0, parms->mirrorPairs, parms->mirrorDstGpu, parms->mirrorSrcGpu, parms->mirrorShiftGpu, &s->gpu, s->gpu.boundary_stream);
}

HaloExchange* initForceHaloExchange(Domain* domain, LinkCell* boxes, int allocDevice)
{
   HaloExchange* hh = initHaloExchangeBase(domain);
   const int* g = boxes->gridSize;
   const int size0 = g[1] * g[2], size1 = (g[0] + 2) * g[2], size2 = (g[0] + 2) * (g[1] + 2);
   const int maxSize = MAXI(size0, MAXI(size1, size2));
   ForceExchangeParms* parms = (ForceExchangeParms*)calloc(1, sizeof(ForceExchangeParms));
   parms->capacityAtoms = maxSize * boxes->maxAtoms;
   hh->bufCapacity = parms->capacityAtoms * (int)sizeof(real_t);
   hh->loadBuffer = loadForceBuffer;
   hh->msgBytes = forceMsgBytes;
   hh->msgBytes2 = forceMsgBytes2;
   hh->unloadBuffer = unloadForceBuffer;
   hh->destroy = destroyForceExchange;
   parms->nCells[HALO_X_MINUS] = parms->nCells[HALO_X_PLUS] = size0;
   parms->nCells[HALO_Y_MINUS] = parms->nCells[HALO_Y_PLUS] = size1;
   parms->nCells[HALO_Z_MINUS] = parms->nCells[HALO_Z_PLUS] = size2;
   for (int f = 0; f < 6; ++f) {
      parms->sendCells[f] = mkForceSendCellList(boxes, f, parms->nCells[f]);
      parms->recvCells[f] = mkForceRecvCellList(boxes, f, parms->nCells[f]);
   }
   hh->type = 1;
   hh->parms = parms;
   hh->deviceBuffers = allocDevice;
   hh->nTotalBoxes = boxes->nTotalBoxes; if (allocDevice) hh->mirror = slotMirror;
   hh->msgHeaderBytes = 0; hh->msgBytesPerAtom = (int)sizeof(real_t); hh->capacityAtoms = parms->capacityAtoms;
   hh->countPtrs = forceCountPtrs; hh->setBounds = forceSetBounds;
   if (allocDevice) { hh->loadBuffer2 = loadSlotBuffer2; hh->unloadBuffer2 = unloadSlotBuffer2; }
   if (allocDevice) {
      for (int f = 0; f < 6; ++f) {
         parms->sendCellsGpu[f] = uploadInts(parms->sendCells[f], parms->nCells[f]);
         parms->recvCellsGpu[f] = uploadInts(parms->recvCells[f], parms->nCells[f]);
         parms->sendOffsetsGpu[f] = (int*)comdDeviceMalloc((long)(parms->nCells[f] + 1) * sizeof(int));
         parms->recvOffsetsGpu[f] = (int*)comdDeviceMalloc((long)(parms->nCells[f] + 1) * sizeof(int));
      }
      parms->d_cellOffsets = (int*)comdDeviceMalloc((long)(maxSize + 1) * sizeof(int));
      hh->sendBufM = (char*)comdDeviceMalloc(hh->bufCapacity); hh->sendBufP = (char*)comdDeviceMalloc(hh->bufCapacity);
      hh->recvBufM = (char*)comdDeviceMalloc(hh->bufCapacity); hh->recvBufP = (char*)comdDeviceMalloc(hh->bufCapacity);
   }
   return hh;
}

/* ---- position refresh plugin (Verlet-list mode) ----------------------------------------------------------------
 * Between list builds nothing is re-binned, so the halo cells of a rank hold the same atoms, in the same slots, as the send
 * cells of its neighbour did at the build: the exchange is positional, exactly like dF/drho, with three reals per atom and
 * the periodic shift of the face added on the way out.  Cell lists, growing x -> y -> z footprint and buffers are those of
 * initForceHaloExchange.  (The reference re-sends whole atoms and looks their slots up in a gid hash table,
 * haloExchange.c:1622-1700.)  Message sizes cannot change between builds: read once, then cached. */
static int loadPositionBuffer(void* vparms, void* vdata, int face, char* buf)
{
   ForceExchangeParms* parms = (ForceExchangeParms*)vparms;
   SimFlat* s = (SimFlat*)vdata;
   s->gpu.msgBoundAtoms = parms->sendBound[face];
   loadPositionBufferFromGpu((real_t*)buf, parms->nCells[face], parms->sendCellsGpu[face], parms->sendOffsetsGpu[face], parms->shift[face],
                             &s->gpu, s->gpu.boundary_stream);
   s->gpu.msgBoundAtoms = 0;
   return parms->msgBytesCached[face];
}

static int positionMsgBytes(void* vparms, void* vdata, int face, char* buf)
{
   (void)buf;
   ForceExchangeParms* parms = (ForceExchangeParms*)vparms;
   SimFlat* s = (SimFlat*)vdata;
   if (parms->msgBytesCached[face] < 0) {
      int n = comdReadDeviceInt(parms->sendOffsetsGpu[face] + parms->nCells[face], s->gpu.boundary_stream);
      parms->msgBytesCached[face] = 3 * n * (int)sizeof(real_t);
   }
   return parms->msgBytesCached[face];
}

static void unloadPositionBuffer(void* vparms, void* vdata, int face, int bufSize, char* buf)
{
   ForceExchangeParms* parms = (ForceExchangeParms*)vparms;
   SimFlat* s = (SimFlat*)vdata;
   (void)bufSize;
   unloadPositionBufferToGpu((const real_t*)buf, parms->nCells[face], parms->recvCellsGpu[face], parms->recvOffsetsGpu[face], &s->gpu, s->gpu.boundary_stream);
}

HaloExchange* initPositionHaloExchange(Domain* domain, LinkCell* boxes, int allocDevice)
{
   /* same cell lists and offsets as the force exchange; 3x the payload */
   HaloExchange* hh = initHaloExchangeBase(domain);
   const int* g = boxes->gridSize;
   const int size0 = g[1] * g[2], size1 = (g[0] + 2) * g[2], size2 = (g[0] + 2) * (g[1] + 2);
   const int maxSize = MAXI(size0, MAXI(size1, size2));
   ForceExchangeParms* parms = (ForceExchangeParms*)calloc(1, sizeof(ForceExchangeParms));
   parms->positions = 1;
   parms->capacityAtoms = maxSize * boxes->maxAtoms;
   hh->bufCapacity = 3 * parms->capacityAtoms * (int)sizeof(real_t);
   hh->loadBuffer = loadPositionBuffer;
   hh->msgBytes = positionMsgBytes;
   hh->unloadBuffer = unloadPositionBuffer;
   hh->destroy = destroyForceExchange;
   parms->nCells[HALO_X_MINUS] = parms->nCells[HALO_X_PLUS] = size0;
   parms->nCells[HALO_Y_MINUS] = parms->nCells[HALO_Y_PLUS] = size1;
   parms->nCells[HALO_Z_MINUS] = parms->nCells[HALO_Z_PLUS] = size2;
   for (int f = 0; f < 6; ++f) {
      parms->sendCells[f] = mkForceSendCellList(boxes, f, parms->nCells[f]);
      parms->recvCells[f] = mkForceRecvCellList(boxes, f, parms->nCells[f]);
      parms->msgBytesCached[f] = -1;
   }
   for (int a = 0; a < 3; ++a) {               /* haloExchange.c:316-323 */
      if (domain->procCoord[a] == 0)                       parms->shift[2*a][a]     = +1.0 * domain->globalExtent[a];
      if (domain->procCoord[a] == domain->procGrid[a] - 1) parms->shift[2*a + 1][a] = -1.0 * domain->globalExtent[a];
   }
   hh->type = 2;
   hh->parms = parms;
   hh->deviceBuffers = allocDevice;
   hh->nTotalBoxes = boxes->nTotalBoxes; if (allocDevice) hh->mirror = slotMirror;
   hh->msgHeaderBytes = 0; hh->msgBytesPerAtom = 3 * (int)sizeof(real_t); hh->capacityAtoms = parms->capacityAtoms;
   hh->countPtrs = forceCountPtrs; hh->setBounds = forceSetBounds;
   if (allocDevice) { hh->loadBuffer2 = loadSlotBuffer2; hh->unloadBuffer2 = unloadSlotBuffer2; }
   hh->exactCounts = 1;                      /* slots are frozen between list builds: the counts of the first exchange hold until the next build */
   if (allocDevice) {
      for (int f = 0; f < 6; ++f) {
         parms->sendCellsGpu[f] = uploadInts(parms->sendCells[f], parms->nCells[f]);
         parms->recvCellsGpu[f] = uploadInts(parms->recvCells[f], parms->nCells[f]);
         parms->sendOffsetsGpu[f] = (int*)comdDeviceMalloc((long)(parms->nCells[f] + 1) * sizeof(int));
         parms->recvOffsetsGpu[f] = (int*)comdDeviceMalloc((long)(parms->nCells[f] + 1) * sizeof(int));
      }
      parms->d_cellOffsets = (int*)comdDeviceMalloc((long)(maxSize + 1) * sizeof(int));
      hh->sendBufM = (char*)comdDeviceMalloc(hh->bufCapacity); hh->sendBufP = (char*)comdDeviceMalloc(hh->bufCapacity);
      hh->recvBufM = (char*)comdDeviceMalloc(hh->bufCapacity); hh->recvBufP = (char*)comdDeviceMalloc(hh->bufCapacity);
   }
   return hh;
}

/* after a list build: occupancies are frozen until the next one -- scan the twelve cell lists once, forget the cached sizes */
void preparePositionExchange(HaloExchange* hh, SimFlat* sim)
{
   ForceExchangeParms* parms = (ForceExchangeParms*)hh->parms;
   prepareForceExchange(hh, sim);
   for (int f = 0; f < 6; ++f) parms->msgBytesCached[f] = -1;
   invalidateHaloSizes(hh);
}

void destroyHaloExchange(HaloExchange** pp)
{
   if (!pp || !*pp) return;
   HaloExchange* hh = *pp;
   hh->destroy(hh->parms);
   free(hh->parms);
   if (hh->deviceBuffers) {
      comdDeviceFree(hh->sendBufM); comdDeviceFree(hh->sendBufP); comdDeviceFree(hh->recvBufM); comdDeviceFree(hh->recvBufP);
   }
   for (int a = 0; a < 3; ++a) {
      if (hh->spec[a].event) { if (hh->spec[a].pending) comdEventSynchronize(hh->spec[a].event); comdEventDestroy(hh->spec[a].event); }
      if (hh->spec[a].mirror) comdHostFreePinned(hh->spec[a].mirror);
   }
   free(hh);
   *pp = NULL;
}

/* Occupancies are final once the atom exchange has been sorted: scan all twelve force-exchange cell lists in one launch. */
void prepareForceExchange(HaloExchange* hh, SimFlat* sim)
{
   ForceExchangeParms* parms = (ForceExchangeParms*)hh->parms;
   if (haloMirrorFirstAxis(hh) == 0) return;                 /* no message is packed: no offsets needed */
   int* lists[12]; int* offs[12]; int n[12];
   for (int f = 0; f < 6; ++f) {
      lists[f] = parms->sendCellsGpu[f];     offs[f] = parms->sendOffsetsGpu[f];     n[f] = parms->nCells[f];
      lists[6 + f] = parms->recvCellsGpu[f]; offs[6 + f] = parms->recvOffsetsGpu[f]; n[6 + f] = parms->nCells[f];
   }
   scanCellListsGpu(&sim->gpu, 12, lists, n, offs, sim->gpu.boundary_stream);
}

/* ---- driver -------------------------------------------------------------------------------------------------- */
void invalidateHaloSizes(HaloExchange* hh) { for (int a = 0; a < 3; ++a) { hh->spec[a].valid = 0; hh->spec[a].haveBound = 0; } }

static int handshakeCached = -1;
static int handshakeForced(void)
{
   if (handshakeCached < 0) { const char* e = getenv("COMD_HALO_HANDSHAKE"); handshakeCached = e && atoi(e) != 0; }
   return handshakeCached;
}
/* bench.py's self-check: the same steps again with exact sizes swapped before every exchange, in the same process (1 on, 0 off, -1 back to the environment) */
void comdSetHaloHandshake(int on) { handshakeCached = on; }

/* transfer size both ends derive from last step's count: + 12.5 % + 64 atoms.  COMD_HALO_SLACK="percent,atoms" replaces the two numbers
 * (the tests set 0,0 to see a message outgrow its bound and the run stop) */
static int boundOf(int last, int capacityAtoms)
{
   static int pct = -1, atoms = 64;
   if (pct < 0) {
      pct = 1250;                                          /* hundredths of a percent */
      const char* e = getenv("COMD_HALO_SLACK");
      double p = 0.0; int a = 0;
      if (e && sscanf(e, "%lf,%d", &p, &a) == 2 && p >= 0.0 && a >= 0) { pct = (int)(p * 100.0 + 0.5); atoms = a; }
   }
   long b = (long)last + (long)last * pct / 10000 + atoms;
   return b > capacityAtoms ? capacityAtoms : (int)b;
}

/* A message that used more than half of its slack in ONE step is growing fast (a shock front, a melting surface): its next transfer gets four times the
 * slack -- + 50 % + 256 atoms -- before the run would have to stop a step later.  Decided per MESSAGE from two numbers both of its ends hold
 * (this count and the count its last bound was derived from), so sender and receiver still agree without talking; ranks with three or more neighbours
 * on an axis need exactly that -- a per-rank switch to the handshake would not be symmetric. */
static int boundGrowing(int now, int before, int lastBound, int capacityAtoms)
{
   const int normal = boundOf(now, capacityAtoms);
   if (2L * (now - before) <= (long)lastBound - before) return normal;
   long wide = (long)now + now / 2 + 256;
   if (wide < normal) wide = normal;
   return wide > capacityAtoms ? capacityAtoms : (int)wide;
}

void exchangeData(HaloExchange* hh, void* data, int iAxis)
{
   const int faceM = 2 * iAxis, faceP = faceM + 1;
   const int nbrM = hh->nbrRank[faceM], nbrP = hh->nbrRank[faceP];
   const int selfOnly = nbrM == getMyRank() && nbrP == getMyRank() && !loopbackParallel();
   HaloSpec* sp = &hh->spec[iAxis];
   const int sized = !selfOnly && hh->deviceBuffers && hh->countPtrs && sizedExchangeAvailable() && !handshakeForced();
   const int useSized = sized && sp->valid;
   int bound[4] = { 0, 0, 0, 0 };
   if (useSized) {
      if (sp->pending) { comdEventSynchronize(sp->event); sp->pending = 0; }      /* recorded a whole step ago */
      /* the previous exchange of this axis was posted with lastBound[]: a message that turned out larger was cut short (its pack kernel refused
       * it and raised the device flag).  Both ends see the same count, so both stop here, before any later transfer is sized from diverged cells. */
      for (int i = 0; sp->haveBound && i < 4; ++i)
         if (sp->mirror[i] > sp->lastBound[i]) {
            fprintf(stderr, "Rank %d: a halo message overflowed its buffer, or grew by more than 12.5 %% + 64 atoms in one step (%d atoms against an agreed %d; "
                            "exchange %d, axis %d).  COMD_HALO_HANDSHAKE=1 exchanges exact sizes.\n", getMyRank(), sp->mirror[i], sp->lastBound[i], hh->type, iAxis);
            exit(-1);
         }
      for (int i = 0; i < 4; ++i)
         bound[i] = hh->exactCounts ? sp->mirror[i] : sp->haveBound ? boundGrowing(sp->mirror[i], sp->lastCount[i], sp->lastBound[i], hh->capacityAtoms)
                                                                    : boundOf(sp->mirror[i], hh->capacityAtoms);
      for (int i = 0; i < 4; ++i) { sp->lastBound[i] = bound[i]; sp->lastCount[i] = sp->mirror[i]; }
      sp->haveBound = 1;
      if (hh->setBounds) { hh->setBounds(hh->parms, faceM, bound[0], bound[3]); hh->setBounds(hh->parms, faceP, bound[1], bound[2]); }
   }
   int nSendM, nSendP;
   if (hh->loadBuffer2) {                                  /* both faces: one scan launch, one pack launch */
      int n[2];
      hh->loadBuffer2(hh->parms, data, faceM, hh->sendBufM, faceP, hh->sendBufP, n);
      nSendM = n[0]; nSendP = n[1];
   } else {
      nSendM = hh->loadBuffer(hh->parms, data, faceM, hh->sendBufM);
      nSendP = hh->loadBuffer(hh->parms, data, faceP, hh->sendBufP);
   }

   if (selfOnly) {
      /* this rank is its own neighbour along the axis: what it sends through the minus face arrives through its plus
       * face.  Unpack straight from the send buffers (the reference's comm path has the same shortcut, haloExchange.c:788-853). */
      if (hh->unloadBuffer2) hh->unloadBuffer2(hh->parms, data, faceM, nSendP, hh->sendBufP, faceP, nSendM, hh->sendBufM);
      else {
         hh->unloadBuffer(hh->parms, data, faceM, nSendP, hh->sendBufP);
         hh->unloadBuffer(hh->parms, data, faceP, nSendM, hh->sendBufM);
      }
      return;
   }
   if (useSized) {
      /* sizes agreed beforehand: four transfers enqueued, no size exchange, no host synchronisation; true counts are on the device */
      SimFlat* sim = (SimFlat*)data;
      const int h = hh->msgHeaderBytes, b = hh->msgBytesPerAtom;
      sendReceiveDevice2Sized(hh->sendBufM, h + bound[0] * b, nbrM, hh->recvBufP, h + bound[2] * b,
                              hh->sendBufP, h + bound[1] * b, nbrP, hh->recvBufM, h + bound[3] * b, sim->gpu.boundary_stream);
      if (hh->unloadBuffer2) hh->unloadBuffer2(hh->parms, data, faceM, -1, hh->recvBufM, faceP, -1, hh->recvBufP);
      else {
         hh->unloadBuffer(hh->parms, data, faceM, -1, hh->recvBufM);
         hh->unloadBuffer(hh->parms, data, faceP, -1, hh->recvBufP);
      }
      if (hh->setBounds) { hh->setBounds(hh->parms, faceM, 0, 0); hh->setBounds(hh->parms, faceP, 0, 0); }
   } else {
      if (nSendM < 0 && nSendP < 0 && hh->msgBytes2) {
         int n[2];
         hh->msgBytes2(hh->parms, data, faceM, hh->sendBufM, faceP, hh->sendBufP, n);
         nSendM = n[0]; nSendP = n[1];
      }
      if (nSendM < 0 && hh->msgBytes) nSendM = hh->msgBytes(hh->parms, data, faceM, hh->sendBufM);
      if (nSendP < 0 && hh->msgBytes) nSendP = hh->msgBytes(hh->parms, data, faceP, hh->sendBufP);
      int nRecvP, nRecvM;
      if (hh->deviceBuffers) {
         SimFlat* sim = (SimFlat*)data;
         int nRecv[2];
         sendReceiveDevice2(hh->sendBufM, nSendM, nbrM, hh->recvBufP, hh->sendBufP, nSendP, nbrP, hh->recvBufM, hh->bufCapacity,
                            sim->gpu.boundary_stream, nRecv);
         nRecvP = nRecv[0]; nRecvM = nRecv[1];
      } else {
         nRecvP = sendReceiveParallel(hh->sendBufM, nSendM, nbrM, hh->recvBufP, hh->bufCapacity, nbrP);
         nRecvM = sendReceiveParallel(hh->sendBufP, nSendP, nbrP, hh->recvBufM, hh->bufCapacity, nbrM);
      }
      if (hh->unloadBuffer2) hh->unloadBuffer2(hh->parms, data, faceM, nRecvM, hh->recvBufM, faceP, nRecvP, hh->recvBufP);
      else {
         hh->unloadBuffer(hh->parms, data, faceM, nRecvM, hh->recvBufM);
         hh->unloadBuffer(hh->parms, data, faceP, nRecvP, hh->recvBufP);
      }
   }
   if (sized && !(hh->exactCounts && useSized)) {
      /* leave the four counts of this phase where the next exchange of the axis finds them without asking the device */
      SimFlat* sim = (SimFlat*)data;
      const int* p[4];
      hh->countPtrs(hh->parms, hh, faceM, faceP, p);
      if (!sp->mirror) { sp->mirror = (int*)comdHostMallocPinned(4 * sizeof(int)); sp->event = comdEventCreate(); }
      comdMirrorCounts(p[0], p[1], p[2], p[3], sp->mirror, sim->gpu.boundary_stream);
      comdEventRecord(sp->event, sim->gpu.boundary_stream);
      sp->pending = 1; sp->valid = 1;
   }
}

void haloExchange(HaloExchange* hh, void* data)
{
   startTimer(commHaloTimer);
   const int first = haloMirrorFirstAxis(hh);               /* axes first..2: this rank is its own neighbour and the plugin can mirror cells directly */
   for (int iAxis = 0; iAxis < first; ++iAxis) exchangeData(hh, data, iAxis);
   if (first < 3) hh->mirror(hh, data, first);
   stopTimer(commHaloTimer);
}
