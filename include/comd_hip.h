/* comd_hip.h -- C ABI of libcomd_hip.so, the MI355X (gfx950) device library behind CoMD's
 * pair-force + velocity-Verlet hot path.
 *
 * Every entry point replaces one extern "C" launch wrapper of the reference
 * (e-ago/CoMD-CUDA-Async, src-mpi/gpu_kernels.h:13-92 and src-mpi/gpu_utility.h:55-69); the
 * comment above each declaration cites the interface it stands in for.  Signatures carry plain
 * pointers, sizes and PODs only -- no HIP, torch or C++ types -- so a C host (ours, or the
 * reference's src-mpi C files with the shim of INTEGRATION.md) links against it directly.
 *
 * Differences from the reference ABI, all deliberate:
 *  - MAXATOMS (a -D macro in the reference Makefile:16) is the run-time field SimGpu.maxAtoms.
 *  - there is no a_list/i_list/b_list indirection (gpu_types.h:115-121): kernels are launched over
 *    cell slots (cell*maxAtoms + i) and mask by nAtoms[cell].
 *  - every cell is kept in ascending-gid order on the device, so atom order is a pure function of
 *    cell membership (the reference sorts boundary/halo cells only, gpu_kernels.cu:1013-1043).
 *  - entry points that took SimFlat* take SimGpu* (+ explicit scalars): the device library does
 *    not know the host's simulation struct.
 *  - streams are opaque `comdStream_t` (a hipStream_t underneath; NULL = the default stream).
 *  - errors: any HIP failure prints "Rank r, GPU g, Error in file f at line l" + the HIP error
 *    string to stderr and exit(-1)s, as CUDA_CHECK does (gpu_utility.h:71-90).
 */
#ifndef COMD_HIP_H
#define COMD_HIP_H

#ifdef __cplusplus
extern "C" {
#endif

/* mytype.h:8-21: one build per precision.  -DCOMD_SINGLE (make PRECISION=single -> libcomd_hip_sp.so / libcomd_host_sp.so) makes every
 * position, momentum, force, energy, table sample and message field a float, as the reference's DOUBLE_PRECISION = OFF build does. */
/* (next to the reference's own mytype.h / defines.h -- a reference host file compiled against include/comd_hip_shim.h -- their definitions of the same
 * names stand: same types, same values) */
#ifndef __MYTYPE_H_
#ifdef COMD_SINGLE
typedef float real_t;
#else
typedef double real_t;                 /* mytype.h:16 (COMD_DOUBLE build) */
#endif
typedef struct vec_t { real_t* x; real_t* y; real_t* z; } vec_t;          /* mytype.h:24-28 */
#endif
typedef void*  comdStream_t;

/* force-kernel variants, defines.h:11-17 (warp_atom runs as thread_atom, warp_atom_nl as thread_atom_nl) */
#ifndef __DEFINES_H_
enum { THREAD_ATOM = 0, THREAD_ATOM_NL = 1, WARP_ATOM = 2, WARP_ATOM_NL = 3, CTA_CELL = 4, CPU_NL = 5 };
#endif

/* gpu_types.h:48-58 */
typedef struct InterpolationObjectGpuSt {
   int     n;
   real_t  x0, xn, invDx, invDxHalf, invDxXx0;
   real_t* values;                     /* device, n+3 entries, values[0] is the leading pad */
} InterpolationObjectGpu;

/* gpu_types.h:72-79 */
typedef struct LjPotentialGpuSt {
   real_t cutoff, sigma, epsilon;
   /* thread_atom: before each force evaluation every full wave (64 consecutive slots of a cell) gets the list of the stencil atoms that lie
    * within the cutoff of the bounding box of its atoms; the force kernel tests only those.  Allocated by the first thread_atom launch. */
   unsigned* waveCand;                 /* device [nLocalBoxes * waveCandWaves][waveCandCap] global slots */
   int*      waveCandCount;            /* device [nLocalBoxes * waveCandWaves][2]: {own-cell candidates, all candidates}; second < 0: no list */
   int       waveCandCap, waveCandWaves;
   int       packedCap;                /* slots per cell in packedR: the fullest cell seen at allocation + 16, at most maxAtoms; -1: no lists (the array would pass 4 GiB) */
   real_t*   packedR[2];               /* device [nTotalBoxes * packedCap][4]: {x, y, z, cutoff^2} of the occupied slots, refreshed by every thread_atom
                                        * force call; [1] is used by calls on a stream other than interior_stream when the force is split (-a 1) */
   float*    packedF[2];               /* device [nTotalBoxes * packedCap][4]: the same positions in single precision, relative to the corner of the local domain:
                                        * what the list build tests against its boxes (16 bytes per stencil atom instead of 24, fp32 arithmetic) */
} LjPotentialGpu;

/* gpu_types.h:60-69: cubic spline in r^2 (-P, `spline` argument of eamForce*Gpu): coefficients {a,b,c,d} per table interval,
 * f(r) = ((a r2 + b) r2 + c) r2 + d; the interval is picked with single-precision arithmetic as in the reference */
typedef struct InterpolationSplineObjectGpuSt {
   int     n;
   float   x0, xn, invDx, invDxXx0;
   real_t* coefficients;               /* device, 4 * n */
} InterpolationSplineObjectGpu;

/* gpu_types.h:81-96 */
typedef struct EamPotentialGpuSt {
   real_t cutoff;
   InterpolationObjectGpu phi, rho, f;
   InterpolationSplineObjectGpu phiS, rhoS;   /* allocated when GpuConfig.phiSpline / rhoSpline are given; F(rhobar) stays quadratic (gpu_utility.c:443) */
   real_t* rhobar;                     /* device [nTotalBoxes*maxAtoms] */
   real_t* dfEmbed;                    /* device [nTotalBoxes*maxAtoms] */
   /* cta_cell: pass 1 leaves each atom's in-cutoff neighbours (16-bit numbers in the staging order of its cell's stencil) here and pass 3
    * reads them back instead of testing the stencil again; laid out the way pass 3's lanes consume them -- per atom [16 lanes][8 trips]
    * words of two numbers -- so that a lane fetches its share with two 16-byte loads; allocated by the first cta_cell launch */
   unsigned* pairRows;                 /* device [nLocalBoxes][rounds of 16 atoms][2 quads][64 lanes] uint4 (round 3: [nLocalBoxes*maxAtoms][128] words; COMD_EAM_CTA=cell still uses that) */
   unsigned short* pairRowCount;       /* device [nLocalBoxes*maxAtoms] */
   int     pairRowLen;                 /* neighbours a row can hold (<= 256) */
   /* cta_cell, brick form (round 3, hip/eam_brick_kernels.h): the selection marks of launches that cover a cell list (the cells with
    * cellSel[c] == the launch's tag); allocated on first use.  The rows above keep their layout, the numbers now index the LDS image of the atom's brick. */
   int     brickImageCap;              /* host: records the LDS image of a brick holds, fixed by the first brick launch from the occupancies it finds */
   int*    brickGroup;                 /* device [nLocalBoxes]: 1 = the cell's brick holds a boundary cell, 2 = it does not: the overlap mode's boundary / interior launches of cta_cell
                                        * take whole bricks (a brick with cells of both lists would be staged twice per pass); built by the first launch over boundary_cells or interior_cells */
   int     brickGroupBy, brickGroupBz; /* the brick shape the groups were built for */
   int*    brickList;                  /* device [2][brickListStride]: the bricks of group 1, of group 2 (what a group launch's workgroups take) */
   int     brickCount[2], brickListStride;
   int     brickListMakes;             /* host: times the brick lists were made (eamBrickBuildLists) */
   int     brickListsValid;            /* host: the lists below were made for occupancies that still fit (remade when a build finds a brick that does not) */
   int     brickCountAll;              /* Verlet rows (slabFormat 4): the lists are made by every list build -- [0] boundary bricks, [1] the others, [2 * stride ..) all; an entry is a
                                        * brick number, or with bits 28-29 set the lower (1) / upper (2) z half of a brick whose block would outgrow the LDS image */
   int*    cellSel;                    /* device [nLocalBoxes] */
   int     selTag;                     /* host: tag of the last list launch */
   /* [round 4] per cell, the selection (64-bit mask of the brick's cells) its brick was staged for when pass 1 wrote the cell's rows; pass 3 of the same force evaluation must be launched over
    * the same partition of the cells -- the 16-bit numbers in pairRows index an image whose composition depends on it -- and raises status[3] bit 2 otherwise */
   unsigned long long* brickSel;       /* device [nLocalBoxes] */
   int*    brickStats;                 /* device [2]: {longest Verlet row of the last list build, bricks that took the thread-per-atom form since the last comdEamBrickStats} */
   int     brickBy, brickBz;           /* host: the brick shape of this simulation (fixed by the first launch: rows index the image of that shape) */
   /* [round 4] method thread_atom (EAM_Force_atom_brick, hip/eam_atom_brick_kernels.h): thread per atom inside a brick workgroup; a shape and an image of its own */
   int     atomBrickBy, atomBrickBz;   /* host: brick shape (0: chosen by the first launch) */
   int     atomBrickImageCap;          /* host: records of its LDS image (0: sized by the next launch) */
   unsigned* atomRows;                 /* device [bricks][rows / 8][256] uint4: the rows pass 1 leaves for pass 3, 8 image numbers per element, atom = its index in the brick */
   unsigned short* atomRowCount;       /* device [bricks][256]: their lengths (0xffff: no row) */
   unsigned long long* atomBrickSel;   /* device [nLocalBoxes]: the selection of its brick a cell's rows were numbered under (0: none) */
   int     atomRowsValid;              /* host: the last pass 1 of this simulation was EAM_Force_atom_brick's (else pass 3 tests again) */
} EamPotentialGpu;

/* gpu_types.h:98-112 */
typedef struct LinkCellGpuSt {
   int    nLocalBoxes, nTotalBoxes;
   int    gridSize[3];
   real_t localMin[3], localMax[3], invBoxSize[3];
   int*   nAtoms;                      /* device [nTotalBoxes] */
   int*   boxIDLookUp;                 /* device [nLocalBoxes] or NULL: -H renumbering of the local cells, tuple index -> id (gpu_types.h:114) */
   int*   boxIDLookUpReverse;          /* device [nLocalBoxes] or NULL: id -> tuple index (gpu_types.h:115 stores an int3) */
} LinkCellGpu;

/* gpu_types.h:120-146 NeighborListGpu.  Verlet lists for the *_nl methods: every atom within cutoff + skinDistance of a local
 * atom, valid until some atom has moved more than skinDistance/2 since the build.  An entry is the neighbour's global slot:
 * between builds no atom changes slot (nothing is re-binned, halo copies are refreshed in place). */
typedef struct NeighborListGpuSt {
   int*   list;                        /* device [nLocalBoxes * maxNeighbors * maxAtoms]: entry k of atom i of cell c at (c*maxNeighbors + k)*maxAtoms + i */
   int*   nNeighbors;                  /* device [nLocalBoxes * maxAtoms] */
   int    maxNeighbors;                /* rows per cell (gpu_neighborList.c:50 MAXNEIGHBORLISTSIZE) */
   vec_t  lastR;                       /* device [nLocalBoxes * maxAtoms]: positions at the last build */
   int*   updateRequired;              /* device view of updateRequiredHost */
   int*   updateRequiredHost;          /* pinned host [16]: [0] raised by NeighborListUpdateRequired (the stand-alone test), cleared by the host; [1] / [2] the number of the last
                                        * fused drift kernel that saw an atom beyond skin/2 / beyond the soft threshold since -- never cleared, compared with buildDrift;
                                        * [3] progress: drift kernel G writes G - 1 as it starts */
   int    checkFused;                  /* host: a fused drift kernel has tested the positions it wrote since the last decision */
   int    driftCount, buildDrift;      /* host: fused drift kernels launched so far; the count when the lists were last built */
   int    lastInterval;                /* host: drifts the previous lists lasted (0: unknown) -- sizes the margin of the deferred decision */
   real_t softHalf2;                   /* host: (skin/2 - margin)^2 of the current lists */
   real_t skinDistance, skinDistance2, skinDistanceHalf2;
   /* LJ (cells of <= 512 slots): the list is kept per group of stencil cells (the 3 x-planes of 9 cells) as 16-bit indices into the
    * LDS staging of that group: list16[((c*3 + g)*slabRows + k)*maxAtoms + i], nNeighbors[(c*3 + g)*maxAtoms + i]; `list` is unused */
   int    slabFormat;                  /* 0: `list` of global slots; 1: LJ slab lists; 2: EAM, 16-bit record numbers into a wave's staging of the
                                        * whole 27-cell stencil: list16[(c*slabRows + k)*maxAtoms + i], nNeighbors[c*maxAtoms + i] (COMD_EAM_NL=lds);
                                        * 4: EAM, rows of the brick kernel (hip/eam_brick_kernels.h, the default since round 4): 16-bit record numbers in the
                                        * LDS image of the atom's brick, per (cell, round of brickRoundAtoms atoms) [brickQuads][64 lanes] 16-byte elements */
   unsigned short* list16;
   int    slabRows;
   int*   stats;                       /* device [2]: {atoms in the largest group, fullest cell} at the last build */
   int    maxSlabAtoms, maxCellAtoms;  /* host copies */
   /* slabFormat 3: pairlists for LJ cta_cell (-L, gpu_lj_cta_cell.h:124-274): no lists, one bit per (wave, 8-neighbour trip of a slab) */
   unsigned* pairlist;                 /* device [nLocalBoxes * pairlistWaves * 9 slabs * 8 words] */
   int    pairlistWaves;               /* waves per cell the bits are laid out for */
   int    pairlistBuildId;             /* host: nBuilds value the bits were generated for (!= nBuilds: the next force call generates) */
   unsigned* brickRows;                /* slabFormat 4: device [nLocalBoxes][brickRounds][brickQuads][64] uint4 */
   unsigned short* brickRowCount;      /* slabFormat 4: device [nLocalBoxes * maxAtoms] list lengths */
   int    brickRowLen, brickRounds, brickQuads, brickRoundAtoms;
   int*   brickStatsMirror;            /* pinned host [2]: {longest row, bricks that did not fit the image} of the last build, copied behind it; the next build looks */
   void*  brickStatsEvent;             /* recorded behind that copy */
   int    forceRebuildFlag;            /* host: the next neighborListUpdateRequiredGpu answers 1 without looking */
   int    nBuilds;                     /* host: builds since AllocateGpu */
} NeighborListGpu;

/* gpu_types.h:148-157 */
typedef struct AtomsGpuSt {
   vec_t   r, p, f;                    /* device SoA, [nTotalBoxes*maxAtoms] each */
   real_t* e;
   int*    iSpecies;
   int*    gid;                        /* -1 marks a hole between updateLinkCellsGpu's two phases */
   NeighborListGpu neighborList;       /* allocated only when GpuConfig.skinDistance > 0 */
} AtomsGpu;

/* gpu_types.h:38-45: the reference's gid -> slot hash table of its list mode.  Counters only here (atoms keep their slots between builds). */
typedef struct HashTableGpuSt { int nMaxEntries, nEntriesPut, nEntriesGet; } HashTableGpu;

/* gpu_types.h:159-190.  Passed by pointer everywhere (the reference passes 856 bytes by value). */
typedef struct SimGpuSt {
   int          maxAtoms;              /* slot capacity of a link cell (Makefile:16 MAXATOMS) */
   int          max_atoms_cell;        /* largest occupancy last seen by updateNAtomsCpu (gpu_types.h:160); 0 = unknown */
   int          deviceId, rank;
   AtomsGpu     atoms;
   int*         neighbor_cells;        /* device [nLocalBoxes*27], self first (gpu_utility.c:520-531) */
   real_t*      species_mass;          /* device [1] */
   real_t       mass;                  /* host copy of species_mass[0] */
   LinkCellGpu  boxes;
   LjPotentialGpu  lj_pot;
   EamPotentialGpu eam_pot;
   int          do_eam;
   /* cell typing for communication overlap, gpu_utility.c:73-163 SetBoundaryCells */
   int          n_boundary_cells, n_interior_cells, n_boundary1_cells;
   int*         boundary_cells;        /* device: local cells within two rings of the surface */
   int*         interior_cells;        /* device: the other local cells */
   int*         boundary1_cells;       /* device: outermost ring */
   comdStream_t boundary_stream, interior_stream;
   /* redistribution scratch (CoMDTypes.h:123-126 flags/tmp_sort) */
   int*         nAtomsPrev;            /* device [nTotalBoxes]: occupancy snapshot */
   int*         cellDirty;             /* device [nTotalBoxes]: membership changed, needs compaction + gid sort */
   int*         cellArrivals;          /* device [3][nTotalBoxes]: atoms mirrorAtomCellsGpu appended to a cell in the phase of axis a, counted beside nAtoms until the next sort */
   int*         status;                /* device [4]: {cell overflow / stencil too large, lost atom, msg overflow or outgrown bound, EAM row / list overflow} */
   real_t*      reduceBuf;             /* device: per-block partial sums for computeEnergy */
   real_t*      pinned;                /* pinned host staging (energies, counts) */
   int          reduceBlocks;
   /* per-simulation switches of the launch wrappers (library globals in round 1) */
   int          needEnergy;            /* comdSetEnergyNeeded: 0 = the coming force evaluations feed no energy read */
   int          forceScansReady;       /* comdForceScansReady: load/unloadForceBuffer*Gpu use the offsets of scanCellListsGpu as they are */
   int          msgBoundAtoms;         /* > 0: load{Force,Position}BufferFromGpu flag status[2] when the listed cells hold more atoms than this
                                        * (the size both ends of a halo message agreed on beforehand, see CommTransport.sendrecv2sized) */
   void*        timing;                /* comdForceTiming*: event pool of this simulation, NULL = off */
   real_t       latticeConstant;       /* GpuConfig.latticeConstant (0: 3.615): density estimate behind the LDS sizing of the cell kernels */
   int          fuseEmbed;             /* host switch, default 0: with method CTA_CELL (and the list method on brick rows, and THREAD_ATOM on the brick image) eamForce1Gpu[Async] also does the work of eamForce2Gpu[Async] for the
                                        * cells it covers (F(rhobar), F'(rhobar) need nothing but the atom's own rhobar) and eamForce2Gpu[Async] returns at
                                        * once -- same e[], dfEmbed[] after the pair of calls, one launch fewer */
   /* scan scratch of the reference-side adapter (include/comd_hip_shim.h comdShimOffsets): the reference's per-face partial_sums arrays hold nCells ints,
    * the scans here need nCells + 1; grown on demand, freed by DestroyGpu; the library itself never touches it */
   int          skipSortAfterUpdate;   /* host switch, default 0: updateLinkCellsGpu leaves the holes of the movers in place; sortAtomsGpu squeezes them.  Only when nothing reads the
                                        * cells in between but mirrorAtomCellsGpu (which skips holes): every axis of the atom exchange mirrored, no overlap mode */
   int*         statusMirrorDev;       /* device view of the four status words' mirror in `pinned`: the fused drift kernels refresh it as they start */
   int          statusMirrored;        /* host: a fused drift kernel has been launched since the last comdPollStatus */
   void*        statusEvent;           /* comdPollStatus: recorded behind the last status mirror (pinned[32..35]), NULL before the first poll */
   int*         adapterScan;
   int          adapterScanCap;
   /* fields the reference's host code assigns (timestep.c:229-236, :329, :350); kept so that those statements compile, not read by the library */
   HashTableGpu d_hashTable;
   int          genPairlist;
   int*         d_updateLinkCellsRequired;   /* device [1], zero: the reference's list mode copies it back to ask "did an atom change cells?" (timestep.c:329) */
} SimGpu;

/* Everything AllocateGpu needs to know about the rank's geometry and potential.
 * Replaces the SimFlat* argument of gpu_utility.c:165-282 AllocateGpu. */
typedef struct GpuConfig {
   int    maxAtoms;
   int    nLocalBoxes, nTotalBoxes, gridSize[3];
   real_t localMin[3], localMax[3], boxSize[3];
   int    do_eam, gpuAsync, rank;
   real_t mass;
   real_t ljCutoff, ljSigma, ljEpsilon;
   real_t eamCutoff;
   int    nPhi, nRho, nF;                         /* table lengths (n, not n+3) */
   real_t phiX0, phiInvDx, rhoX0, rhoInvDx, fX0, fInvDx;
   const real_t *phiValues, *rhoValues, *fValues; /* host, n+3 entries each, element 0 = values[-1] */
   const real_t *phiSpline, *rhoSpline;           /* host, 4*n spline coefficients each (initSplineCoefficients, gpu_utility.c:377-430), or NULL */
   const int* neighborCells;                      /* host [nLocalBoxes*27] */
   const int *boxIDLookUp, *boxIDLookUpReverse;   /* host [nLocalBoxes] each, or NULL (natural cell order) */
   real_t skinDistance;                           /* > 0: allocate Verlet lists (AllocateGpu's third argument, gpu_utility.c:165) */
   int    usePairlist;                            /* with skinDistance > 0: pairlist bits for LJ cta_cell instead of lists (CoMD.c:250-268) */
   int    maxNeighbors;                           /* list rows per atom; 0 = derive from cutoff + skin and the FCC density */
   real_t latticeConstant;                        /* for that estimate */
} GpuConfig;

/* Host-side mirror of the slot arrays (CoMDTypes.h Atoms / gpu_utility.c:432-600 staging). */
typedef struct HostAtoms {
   int*    nAtoms;                     /* [nTotalBoxes] */
   int*    gid;  int* iSpecies;        /* [nTotalBoxes*maxAtoms] */
   real_t *rx, *ry, *rz, *px, *py, *pz, *fx, *fy, *fz, *e;
} HostAtoms;

/* ---- device management: gpu_utility.h:55-69 --------------------------------------------- */
/* SetupGpu(int deviceId), gpu_utility.c:32-71: select the device, print its name. Returns the CU count. */
int  SetupGpu(int deviceId, int rank, int verbose);
/* number of visible devices (CoMD.c:105 cudaGetDeviceCount); 0 when there is none. Never exits. */
int  comdDeviceCount(void);
/* AllocateGpu(SimFlat*, do_eam, skin), gpu_utility.c:165-282 */
void AllocateGpu(SimGpu* sim, const GpuConfig* cfg);
/* SetBoundaryCells(SimFlat*, HaloExchange*), gpu_utility.c:73-163: upload the cell-type lists built by the host */
void SetBoundaryCells(SimGpu* sim, int nBoundary, const int* boundary, int nInterior, const int* interior,
                      int nBoundary1, const int* boundary1);
/* CopyDataToGpu(SimFlat*, do_eam), gpu_utility.c:432-600 */
void CopyDataToGpu(SimGpu* sim, const HostAtoms* host);
/* GetDataFromGpu(SimFlat*), gpu_utility.c:617-653 (f and e included) */
void GetDataFromGpu(SimGpu* sim, HostAtoms* host);
/* updateNAtomsCpu(SimFlat*), gpu_utility.c: refresh the host copy of nAtoms (CoMD.c:445-452 reads it) */
void updateNAtomsCpu(SimGpu* sim, int* nAtomsHost);
/* The staging entry points of gpu_utility.h:60-69 that the reference's cpu_nl path and DEBUG blocks use (timestep.c:309).  Where the reference passes
 * SimFlat*, the device library takes its own SimGpu* and the host arrays as a HostAtoms (as CopyDataToGpu / GetDataFromGpu do). */
void cudaCopyDtH(void* dst, const void* src, int size);                         /* gpu_utility.c:46-49: identical signature */
void GetLocalAtomsFromGpu(SimGpu* sim, HostAtoms* host);                        /* gpu_utility.c:656-673: p, r, gid of the local cells -> host */
void updateGpuHalo(SimGpu* sim, const HostAtoms* host);                         /* gpu_utility.c:714-757: p, r, gid, iSpecies of the halo cells -> device */
void updateNAtomsGpu(SimGpu* sim, const int* nAtomsHost);                       /* gpu_utility.c:602-605 */
/* gpu_utility.c:678-712 (host code there too): the halo cells' atoms as one SoA message (no header) + the scan of their occupancies; returns the atom count */
int  compactHaloCells(const HostAtoms* host, int nLocalBoxes, int nTotalBoxes, int maxAtoms, char* h_compactAtoms, int* h_cellOffset);
/* DestroyGpu(SimFlat*), gpu_utility.c:284-347 */
void DestroyGpu(SimGpu* sim);
/* initLinkCellsGpu(SimFlat*, LinkCellGpu*), gpu_utility.c:757-790: geometry + occupancy array of the device link cells (AllocateGpu calls it) */
void initLinkCellsGpu(LinkCellGpu* boxes, const GpuConfig* cfg);
/* emptyHaloCellsGpu(SimFlat*), gpu_utility.c: zero the halo cells' occupancy */
void emptyHaloCellsGpu(SimGpu* sim, comdStream_t stream);
/* cudaDeviceSynchronize / cudaStreamSynchronize as the host code uses them (eam.c:209, 256) */
void comdDeviceSynchronize(void);
void comdStreamSynchronize(comdStream_t stream);
/* device allocation helpers for buffers the host owns (haloExchange.c:228-246, CoMDTypes.h:125-126) */
void* comdDeviceMalloc(long bytes);
void  comdDeviceFree(void* p);
void* comdHostMallocPinned(long bytes);
void  comdHostFreePinned(void* p);
void  comdMemcpyHtoD(void* dst, const void* src, long bytes);
void  comdMemcpyDtoH(void* dst, const void* src, long bytes);
void  comdMemcpyDtoDAsync(void* dst, const void* src, long bytes, comdStream_t stream);
/* cudaMemcpyAsync / cudaMemset as the reference's host files use them (haloExchange.c:1632, timestep.c:224); kind: 1 H2D, 2 D2H, 3 D2D */
void  comdMemcpyAsync(void* dst, const void* src, long bytes, int kind, comdStream_t stream);
void  comdDeviceMemset(void* p, int value, long bytes);
/* check SimGpu.status; prints and exit(-1)s on cell overflow / lost atoms (DEBUG asserts of gpu_redistribute.h:145-154) */
void comdCheckStatus(SimGpu* sim, const char* where);
/* the same without waiting for the device: every call looks at the status words mirrored into pinned memory by the call BEFORE (an asynchronous
 * 16-byte copy on `stream`, a step old by then) and enqueues the next mirror; a non-zero word goes to comdCheckStatus (which prints and exits).
 * timestep() calls it once per step: a dropped atom or an overflowing cell stops the run one or two steps later, not at the next energy read. */
void comdPollStatus(SimGpu* sim, comdStream_t stream, const char* where);

/* ---- force: gpu_kernels.h:13-24 ------------------------------------------------------------ */
/* ljForceGpu(SimGpu*, interpolation, num_cells, cells_list, plcutoff, method), gpu_kernels.cu:69-122.
 * cells_list (device) == NULL means cells 0..num_cells-1.  interpolation/plcutoff are accepted for
 * signature parity: interpolation must be 0 (table-LJ is out of scope; non-zero exits); plcutoff (cutoff + skin) is what the pairlist
 * bits are generated with when the lists were allocated with GpuConfig.usePairlist and method is CTA_CELL.
 * method THREAD_ATOM_NL / WARP_ATOM_NL walks the Verlet lists (buildNeighborListGpu must have run). */
void ljForceGpu(SimGpu* sim, int interpolation, int num_cells, int* cells_list, real_t plcutoff, int method);
void ljForceGpuAsync(SimGpu* sim, int num_cells, int* cells_list, int method, comdStream_t stream);
/* The per-atom energy array e[] is read only by computeEnergy.  comdSetEnergyNeeded(0) tells the force wrappers that the
 * next evaluations feed no energy read, so they may skip the energy arithmetic; comdSetEnergyNeeded(1) (the default)
 * restores the reference behaviour of computing e[] on every call.  timestep() brackets all but its last step with it. */
void comdSetEnergyNeeded(SimGpu* sim, int on);
/* eamForce{1,2,3}Gpu(SimGpu, method, spline), gpu_kernels.cu:154-249; spline != 0 needs the spline tables (GpuConfig.phiSpline/rhoSpline)
 * and one of the cell methods (THREAD_ATOM, CTA_CELL) */
void eamForce1Gpu(SimGpu* sim, int method, int spline);
void eamForce2Gpu(SimGpu* sim, int method, int spline);
void eamForce3Gpu(SimGpu* sim, int method, int spline);
/* eamForce{1,2,3}GpuAsync(SimGpu, AtomListGpu, num_cells, cells_list, method, stream, spline);
 * the AtomListGpu argument has no counterpart here (no atom lists) */
void eamForce1GpuAsync(SimGpu* sim, int num_cells, int* cells_list, int method, comdStream_t stream, int spline);
void eamForce2GpuAsync(SimGpu* sim, int num_cells, int* cells_list, int method, comdStream_t stream, int spline);
void eamForce3GpuAsync(SimGpu* sim, int num_cells, int* cells_list, int method, comdStream_t stream, int spline);
/* CONTRACT of the cta_cell passes (and of thread_atom_nl with brick rows): eamForce3Gpu[Async] must cover the cells with the same partition as the
 * eamForce1Gpu[Async] calls of the same force evaluation -- all local cells in one call, or boundary_cells and interior_cells (the lists given to
 * SetBoundaryCells; each launch then takes WHOLE bricks of 1 x 4 x 2 cells: every cell of a brick that holds a boundary cell goes with the boundary
 * launch), or the same other cell lists in both passes.  Pass 1 records the selection it staged every brick for and pass 3 compares: a mismatch raises
 * status[3] bit 2 and comdCheckStatus / comdPollStatus stop the run.  (With Verlet rows the numbering does not depend on the selection.)
 * comdEamBrickStats: {bricks that took the thread-per-atom fall-back since the last call (their block outgrew the LDS image), bricks per launch over all
 * cells, records the image holds}; a non-zero first number after a re-size means a density the image cannot hold. */
void comdEamBrickStats(SimGpu* sim, int out[3]);
/* cta_cell: have the next launch read the occupancies again and size the image for the fullest block (call between force evaluations) */
void comdEamBrickResize(SimGpu* sim);
/* what the force wrappers decided for this simulation: {LJ thread_atom candidate lists in use (0: the plain 27-cell walk -- lists did not fit the device memory),
 * records of the EAM brick image, Verlet-list format (NeighborListGpu.slabFormat), cells per EAM brick + 256 x the times the brick lists of the list method were made} */
void comdForcePathInfo(SimGpu* sim, int out[4]);
/* updateNeighborsGpu[Async], gpu_kernels.cu:251-279: the reference materialises 27*MAXATOMS neighbour
 * offsets per cell for its cta_cell/warp_atom EAM kernels; ours gather from the cell table directly,
 * so these are no-ops kept for link compatibility. */
void updateNeighborsGpu(SimGpu* sim, int* temp);
void updateNeighborsGpuAsync(SimGpu* sim, int* temp, int nCells, int* cellList, comdStream_t stream);

/* ---- integrator + energy: gpu_kernels.h:79-83 --------------------------------------------- */
/* advanceVelocityGpu(SimGpu, dt), gpu_kernels.cu:326-334 */
void advanceVelocityGpu(SimGpu* sim, real_t dt);
/* advancePositionGpu(SimGpu*, dt), gpu_kernels.cu:336-349 */
void advancePositionGpu(SimGpu* sim, real_t dt);
/* the first half kick and the drift of a step in one pass over the atoms: bit-identical to advanceVelocityGpu(dtKick)
 * followed by advancePositionGpu(dtDrift) (timestep.c:52-58), one launch and one sweep over p fewer */
void advanceVelocityPositionGpu(SimGpu* sim, real_t dtKick, real_t dtDrift);
/* the second half kick of a step and the first half kick + drift of the next in one pass: bit-identical to advanceVelocityGpu(dtKick1)
 * followed by advanceVelocityPositionGpu(dtKick2, dtDrift) */
void advanceVelocityVelocityPositionGpu(SimGpu* sim, real_t dtKick1, real_t dtKick2, real_t dtDrift);
/* computeEnergy(SimFlat*, real_t eLocal[2]), gpu_kernels.cu:1045-1059: {sum e, sum p^2/2m} of local atoms.
 * Deterministic two-stage reduction (the reference uses fp64 atomics). Blocks until the result is on the host. */
void computeEnergy(SimGpu* sim, real_t* eLocal);

/* ---- redistribute: gpu_kernels.h:84-86 ------------------------------------------------------ */
/* updateLinkCellsGpu(SimFlat*), gpu_kernels.cu:469-504: empty the halo cells, move every local atom whose
 * coordinates left its cell, then compact + gid-sort the cells that changed. */
void updateLinkCellsGpu(SimGpu* sim, comdStream_t stream);
/* buildAtomListGpu(SimFlat*, stream), gpu_kernels.cu:553-570: nothing to rebuild here; kept as the hook
 * where the reference rebuilds a_list after the halo exchange.  No-op. */
void buildAtomListGpu(SimGpu* sim, comdStream_t stream);
/* sortAtomsGpu(SimFlat*, stream), gpu_kernels.cu:1013-1043: compact + gid-sort every cell whose membership
 * changed since the last sort (halo cells, cells that received migrants). */
void sortAtomsGpu(SimGpu* sim, comdStream_t stream);

/* ---- halo pack / unpack: gpu_kernels.h:26-72 ------------------------------------------------ */
/* Atom message, device or host: 16-byte header {int n; int pad[3]} followed by the reference's SoA wire
 * format (gpu_kernels.cu:506-517 getAtomMsgSoAPtr): int gid[n]; int type[n]; real_t rx[n],ry[n],rz[n],px[n],py[n],pz[n]. */
#define COMD_ATOM_MSG_HEADER 16
#define COMD_ATOM_MSG_BYTES_PER_ATOM (8 + 6 * (int)sizeof(real_t))     /* 56 in the double build = sizeof(AtomMsg), haloExchange.h:32-38 */
#ifndef __HALO_EXCHANGE_                /* (next to the reference's haloExchange.h, its definition of the same struct -- the same eight pointers -- stands) */
typedef struct AtomMsgSoASt {          /* haloExchange.h:39-47 AtomMsgSoA */
   int *gid, *type; real_t *rx, *ry, *rz, *px, *py, *pz;
} AtomMsgSoA;
#endif
struct AtomMsgSoASt;
/* getAtomMsgSoAPtr(buffer, &msg, n), gpu_kernels.cu:506-517; buffer points at the header */
void getAtomMsgSoAPtr(char* buffer, struct AtomMsgSoASt* atomMsg, int n);
/* compactCellsGpu(work_d, nCells, d_cellList, sim, d_cellOffsets, d_workScan, shift, stream), gpu_kernels.cu:519-551:
 * exclusive-scan the occupancies of the listed cells and gather their atoms (positions shifted by `shift`) into
 * the message at work_d, in cell-list order.  d_cellOffsets needs nCells+1 ints.  Nothing is copied to the host and
 * the call does not block; the count is in the message header.  capacityAtoms bounds the message. */
void compactCellsGpu(char* work_d, int nCells, int* d_cellList, SimGpu* sim, int* d_cellOffsets,
                     const real_t shift[3], int capacityAtoms, comdStream_t stream);
/* Both faces of an axis phase (index 0 = minus, 1 = plus) with one scan launch and one pack launch / one unpack launch -- the halo
 * driver's path; the single-face entry points above are these with one face.  workP / msgB / bufP == NULL: one face. */
void compactCellsGpu2(char* workM, char* workP, const int nCells[2], int* const d_cellList[2], SimGpu* sim, int* const d_cellOffsets[2],
                      const real_t shiftM[3], const real_t shiftP[3], const int capacityAtoms[2], comdStream_t stream);
void unloadAtomsBufferToGpu2(const char* msgA, int nBufA, int maxAtomsA, const char* msgB, int nBufB, int maxAtomsB, SimGpu* sim, comdStream_t stream);
/* blocking read of a device message's atom count (the reference returns it from compactCellsGpu, :534-535) */
int  atomMsgCountGpu(SimGpu* sim, const char* msg_d, comdStream_t stream);
/* unloadAtomsBufferToGpu(buf, nBuf, SimFlat*, gpu_buf, stream), gpu_kernels.cu:572-617: bin every atom of the device
 * message into its link cell by coordinate (getBoxFromCoord), appending to the cell and marking it for sorting.
 * nBuf < 0: take the count from the message header on the device; maxAtomsInMsg bounds the launch. */
void unloadAtomsBufferToGpu(const char* msg_d, int nBuf, int maxAtomsInMsg, SimGpu* sim, comdStream_t stream);
/* loadForceBufferFromGpu(buf, &nbuf, nCells, cellList, natoms_buf, partial_sums, SimFlat*, gpu_buf, stream), :619-640:
 * gather dfEmbed of the listed cells, in list order, into gpu_buf (real_t[]); d_cellOffsets needs nCells+1 ints.
 * Does not block; the count ends up in d_cellOffsets[nCells]. */
void loadForceBufferFromGpu(real_t* gpu_buf, int nCells, int* d_cellList, int* d_cellOffsets, SimGpu* sim, comdStream_t stream);
/* unloadForceBufferToGpu(buf, nBuf, nCells, cellList, natoms_buf, partial_sums, SimFlat*, gpu_buf, stream), :642-660 */
void unloadForceBufferToGpu(const real_t* gpu_buf, int nCells, int* d_cellList, int* d_cellOffsets, SimGpu* sim, comdStream_t stream);
void loadForceBufferFromGpu2(real_t* bufM, real_t* bufP, const int nCells[2], int* const d_cellList[2], int* const d_cellOffsets[2],
                             const int boundAtoms[2], SimGpu* sim, comdStream_t stream);
void unloadForceBufferToGpu2(const real_t* bufA, const real_t* bufB, const int nCells[2], int* const d_cellList[2], int* const d_cellOffsets[2],
                             SimGpu* sim, comdStream_t stream);
/* The force exchange scans twelve cell lists per step (natoms_buf/partial_sums of haloExchange.c:438-465).  Occupancies are final
 * once the atom exchange has been sorted, so all of them can be scanned in ONE launch: scanCellListsGpu fills d_cellOffsets[i]
 * (nCells[i] + 1 ints) for up to 12 lists; comdForceScansReady(1) then tells load/unloadForceBuffer*Gpu to use those offsets as they are. */
void scanCellListsGpu(SimGpu* sim, int nLists, int** d_cellLists, const int* nCells, int** d_cellOffsets, comdStream_t stream);
void comdForceScansReady(SimGpu* sim, int on);
/* blocking read of one device int (message counts for the multi-rank transport) */
int  comdReadDeviceInt(const int* d_ptr, comdStream_t stream);
/* two device ints behind ONE stream synchronisation (the two message counts of an axis phase) */
void comdReadDeviceInt2(const int* d_a, const int* d_b, int out[2], comdStream_t stream);
/* copy four device ints into pinned host memory (comdHostMallocPinned) from a one-wave kernel on `stream`: no host involvement.
 * The halo driver mirrors the four message counts of an axis phase this way, records an event behind it and reads them a step later. */
void comdMirrorCounts(const int* d0, const int* d1, const int* d2, const int* d3, int* pinnedDst, comdStream_t stream);


/* ---- inter-rank transport: replaces comm.h:40-74 (libmp/GPUDirect-Async) and parallel.h (MPI) --------------
 * A small vtable the host's parallel layer calls.  libcomd_hip.so provides the RCCL-over-xGMI implementation
 * (comdCommInitRank / comdCommInitFromEnv); an embedding program may supply its own (the CPU tests plug
 * torch.distributed/gloo in through callbacks). */
typedef struct CommTransportSt {
   void* ctx;
   /* paired exchange of BYTES (parallel.c:100-118 sendReceiveParallel).  device != 0: the buffers are device memory
    * and the transfer is ordered on `stream`.  Returns the number of bytes received (<= recvCap). */
   int  (*sendrecv)(void* ctx, const void* sendBuf, int sendLen, int dest, void* recvBuf, int recvCap, int source,
                    int device, comdStream_t stream);
   /* optional (may be NULL): both messages of one axis phase in a single call -- send bufM to dstM and bufP to dstP, receive what
    * dstP sends through its minus face into recvP and what dstM sends through its plus face into recvM.  One size handshake and one
    * payload group instead of two of each.  Writes the received byte counts to nRecv[0] (recvP) and nRecv[1] (recvM). */
   void (*sendrecv2)(void* ctx, const void* sendM, int nSendM, int dstM, void* recvP, const void* sendP, int nSendP, int dstP, void* recvM,
                     int recvCap, int device, comdStream_t stream, int nRecv[2]);
   void (*allreduce)(void* ctx, void* buf, int count, int dtype /* 0 int sum, 1 double sum, 2 int max, 3 float sum */);
   void (*bcast)(void* ctx, void* buf, int len, int root);
   void (*barrier)(void* ctx);
   /* optional (may be NULL): the same four transfers as sendrecv2 when BOTH ends of every message already agree on its size in bytes
    * (the halo driver derives the sizes from the counts of the previous step, which sender and receiver both hold; the true count
    * travels in the message).  No size handshake, no host synchronisation: the transfers are simply enqueued on `stream`. */
   void (*sendrecv2sized)(void* ctx, const void* sendM, int nSendM, int dstM, void* recvP, int nRecvP, const void* sendP, int nSendP, int dstP,
                          void* recvM, int nRecvM, int device, comdStream_t stream);
} CommTransport;
#define COMD_UNIQUE_ID_BYTES 128
/* rank 0: create the RCCL unique id (ncclGetUniqueId); the caller distributes the 128 bytes to every rank */
int  comdCommGetUniqueId(char* id128);
/* every rank, after SetupGpu: join the communicator and fill `out` with the RCCL transport */
int  comdCommInitRank(const char* id128, int rank, int nRanks, CommTransport* out);
/* standalone launcher path: RANK / WORLD_SIZE / LOCAL_RANK / MASTER_PORT from the environment, id through a file */
int  comdCommInitFromEnv(CommTransport* out, int* rank, int* nRanks, int* localRank);
void comdCommFinalize(void);
/* what the communicator itself reports: ncclCommCount / ncclCommUserRank / ncclCommCuDevice; returns 0, or -1 when there is none */
int  comdCommInfo(int* nRanks, int* rank, int* device);

/* ---- Verlet neighbour lists (methods thread_atom_nl / warp_atom_nl): gpu_kernels.h:25, 73-78, 87-92 ------------------
 * The reference offers them for EAM only (CoMD.c:291-295 exits for LJ); here both potentials take them. */
/* emptyNeighborListGpu(SimGpu*, boundaryFlag), gpu_kernels.cu:1063-1085: zero the neighbour counts */
void emptyNeighborListGpu(SimGpu* sim, int boundaryFlag);
/* neighborListUpdateRequiredGpu(SimGpu*), gpu_kernels.cu:1449-1484: 1 when forceRebuildFlag is set or some local atom has moved
 * more than skin/2 since the build (blocking read of one flag).  THIS rank's answer; the caller reduces over ranks. */
int  neighborListUpdateRequiredGpu(SimGpu* sim);
/* The same question WITHOUT draining the stream: answered from what the drift kernels up to the one TWO before the last found (every drift kernel reports, as it starts,
 * that its predecessor is done -- a word in pinned memory the host waits on if it ever runs more than a step ahead; no event, no synchronisation: the host stays ahead of
 * the device and kernel launches keep hiding behind the force kernels -- the blocking form costs 0.1 ms per step at EAM 80^3).  To be safe two drifts late the test uses a
 * threshold below skin/2 by the displacement two steps can add, estimated as twice the average growth over the previous lists' life (4 / lastInterval of skin/2, at least
 * 10 %, at most 60 %); the exact rule is still evaluated by every drift kernel, and lists that turn out to have been used beyond skin/2 stop the run with a message
 * (COMD_NL_SYNC=1 selects the blocking form).  First lists of a run, or positions changed by anything but the fused drift kernels: the blocking form. */
int  comdNeighborListUpdateDeferredGpu(SimGpu* sim);
/* neighborListForceRebuildGpu(NeighborListGpu*), gpu_neighborList.c:88-93: the reference's signature (a host function there; an object of the
 * reference that defines it too simply takes precedence over the library's) */
void neighborListForceRebuildGpu(NeighborListGpu* neighborList);
/* buildNeighborListGpu(SimGpu*, method, boundaryFlag), gpu_kernels.cu:1975-2029: list every atom within cutoff + skin of each
 * local atom (cells must be current: call after the atom exchange), snapshot lastR, clear forceRebuildFlag.
 * boundaryFlag is accepted for signature parity (BOTH = 0 is the only mode the reference enables, timestep.c:59-82). */
void buildNeighborListGpu(SimGpu* sim, int method, int boundaryFlag);
/* pairlistUpdateRequiredGpu(SimGpu*), gpu_kernels.cu:1283-1318: same displacement test as the lists (skin/2 since the last build) */
int  pairlistUpdateRequiredGpu(SimGpu* sim);
/* -L: tell the library that a whole force evaluation has run since the last buildNeighborListGpu, i.e. the pairlist bits now exist
 * for every cell (the first ljForceGpu[Async] calls after a build generate them, interior and boundary launches alike) */
void comdPairlistGenerated(SimGpu* sim);
/* gpu_types.h:38-45 / gpu_kernels.h:71,92: the reference's gid -> slot hash table for its list mode.  Not needed here (atoms keep
 * their slots between list builds); the two entry points only keep the counters, so the reference's call sites link and run. */
void initHashTableGpu(HashTableGpu* hashTable, int nMaxEntries);
void emptyHashTableGpu(HashTableGpu* hashTable);
/* comm.h:40-74 (libmp / GPUDirect-Async, excluded from this build): "not in use" answers, so that the reference's host code takes its
 * plain send/receive path; comm_init's arguments (MPI_Comm, gpuId) are ignored */
int  comm_use_comm(void);
int  comm_use_gdrdma(void);
int  comm_use_async(void);
int  comm_use_gpu_comm(void);
int  comm_select_device(int mpiRank);
#ifndef __cplusplus
int  comm_init();                       /* (MPI_Comm comm, int gpuId) at the reference's call site; returns 0 */
#endif
void comm_finalize(void);
/* The rest of the reference's link surface (the 49 + 3 symbols its host objects leave undefined) exists for the libmp /
 * GPUDirect-Async layer only and is reachable only when comm_use_comm() / comm_use_async() answer non-zero, which they never do here.
 * libcomd_hip.so still DEFINES every one of them, so the reference's objects link; calling one prints its name and exit(-1)s:
 *   comm_irecv comm_isend comm_isend_on_stream comm_send_ready comm_send_ready_on_stream comm_wait_ready_on_stream comm_wait_all
 *   comm_wait_all_on_stream comm_flush comm_progress
 *   loadAtomsBufferFromGpu_Async/_Comm unloadAtomsBufferToGpu_Async/_Comm loadForceBufferFromGpu_Async/_Comm
 *   unloadForceBufferToGpu_Async/_Comm unloadForceScanCells exchangeDataForceGpu_KI neighborListUpdateRequiredGpu_Async
 * (no prototypes: their reference signatures carry MPI and libmp types; tests/test_boundary_shim.py walks the list) */
/* Between list builds the halo copies keep their slots and only their positions are refreshed: gather r (+ the periodic shift of
 * the face) of the listed cells, in list order, into gpu_buf (3 real_t per atom); scatter them into the receive cells.  The
 * reference re-sends whole atoms and finds their slots through a gid hash table (haloExchange.c:1622-1700, hashTable.c). */
void loadPositionBufferFromGpu(real_t* gpu_buf, int nCells, int* d_cellList, int* d_cellOffsets, const real_t shift[3], SimGpu* sim, comdStream_t stream);
void unloadPositionBufferToGpu(const real_t* gpu_buf, int nCells, int* d_cellList, int* d_cellOffsets, SimGpu* sim, comdStream_t stream);
void loadPositionBufferFromGpu2(real_t* bufM, real_t* bufP, const int nCells[2], int* const d_cellList[2], int* const d_cellOffsets[2],
                                const int boundAtoms[2], const real_t shiftM[3], const real_t shiftP[3], SimGpu* sim, comdStream_t stream);
void unloadPositionBufferToGpu2(const real_t* bufA, const real_t* bufB, const int nCells[2], int* const d_cellList[2], int* const d_cellOffsets[2],
                                SimGpu* sim, comdStream_t stream);

/* [round 4] Axes on which a rank is its own neighbour (all three on one rank): the halo cells are filled straight from the cells they are images of, one
 * launch for every such axis at the end of the x -> y -> z sequence instead of a pack and an unpack per axis (same values: haloExchange.c:788-853 is the
 * reference's self-neighbour branch, :1504-1520 the ordering the host folds into one source per halo cell).  kind 0: dfEmbed, 1: positions + d_shift[3k..]. */
/* the atom exchange of a self-neighbour axis (both faces): every atom of the send cells is displaced, binned by its coordinates and appended -- pack and
 * unpack of gpu_redistribute.h:376-402, 499-620 in one launch.  firstAxis: the first axis of the mirrored tail (arrivals of the axes firstAxis..axis-1 count as
 * content of a cell); sortAtomsGpu afterwards, as after the message path. */
void mirrorAtomCellsGpu(const int nCells[2], int* const d_cellList[2], const real_t shiftM[3], const real_t shiftP[3], int firstAxis, int axis,
                        SimGpu* sim, comdStream_t stream);
void mirrorSlotCellsGpu(int kind, int nPairs, const int* d_dst, const int* d_src, const real_t* d_shift, SimGpu* sim, comdStream_t stream);

/* ---- device-side timing for bench.py -------------------------------------------------------- */
/* HIP-event pair on a stream: comdEventCreate/Record/ElapsedMs.  Used to time kernels on the stream they
 * run on (torch.cuda.Event only sees torch's own stream). */
void*  comdEventCreate(void);
void   comdEventRecord(void* ev, comdStream_t stream);
float  comdEventElapsedMs(void* start, void* stop);   /* synchronises on `stop` */
void   comdEventSynchronize(void* ev);
void   comdEventDestroy(void* ev);
/* accumulated device time of the force kernels since the last reset: the library brackets every force launch
 * with events when timing is enabled (off by default; adds two event records per launch). */
void   comdForceTimingEnable(SimGpu* sim, int on);
void   comdForceTimingReset(SimGpu* sim);
double comdForceTimingTotalMs(SimGpu* sim, int* nLaunches);
/* ... and of what a force evaluation launches beside them (LJ thread_atom: LJ_PackPositions + LJ_WaveCandidates; the cell marks of an EAM launch
 * over a cell list): one force EVALUATION costs comdForceTimingTotalMs + comdForceTimingAuxMs */
double comdForceTimingAuxMs(SimGpu* sim, int* nLaunches);
/* hipMemGetInfo: free and total device memory in bytes (bench.py sizes its 256^3 leg against it) */
void   comdDeviceMemInfo(long* freeBytes, long* totalBytes);

#ifdef __cplusplus
}
#endif
#endif
