/* comd_host.h -- host side (plain C11) of the MI355X CoMD hot path.
 *
 * Mirrors the reference's operator surface so the loop reads like src-mpi/timestep.c:
 *   BasePotential {force, print, destroy}      CoMDTypes.h:42-53
 *   HaloExchange  {loadBuffer, unloadBuffer, destroy}   haloExchange.h:84-104
 *   timestep(), computeForce(), redistributeAtoms(), kineticEnergyGpu()   timestep.c:48-276
 * and calls the device only through include/comd_hip.h.  No HIP headers here.
 */
#ifndef COMD_HOST_H
#define COMD_HOST_H

#include <stdio.h>
#include <stdint.h>
#include "comd_hip.h"
#include "comd_geometry.h"

#define screenOut stdout

/* mytype.h:10-19: scanf format of a real_t */
#ifdef COMD_SINGLE
#define FMT1 "%g"
#else
#define FMT1 "%lg"
#endif

/* ---- constants.h:14-39 ---- */
#define amuInKilograms  1.660538921e-27
#define fsInSeconds     1.0e-15
#define AngsInMeters    1.0e-10
#define eVInJoules      1.602176565e-19
static const double amuToInternalMass = amuInKilograms * AngsInMeters * AngsInMeters / (fsInSeconds * fsInSeconds * eVInJoules);
static const double kB_eV = 8.6173324e-5;
static const double hartreeToEv = 27.21138505;
static const double bohrToAngs = 0.52917721092;

/* ---- command line: mycommand.h ---- */
typedef struct CommandSt {
   char potDir[1024], potName[1024], potType[1024], method[1024];
   int doeam, nx, ny, nz, xproc, yproc, zproc, nSteps, printRate;
   double dt, lat, temperature, initialDelta, relativeSkinDistance;
   int doHilbert, gpuAsync, gpuProfile, ljInterpolation, spline, usePairlist, maxNeighbors;
   int maxAtoms;          /* extension: link-cell slot capacity, 0 = choose from the lattice (reference: -DMAXATOMS) */
   int quiet;             /* extension: suppress the stdout report (library use) */
   int deviceTimers;      /* extension: HIP-event timing of the phases of timestep() (the reference's host timers are not device-synchronised) */
   double ljCutoffSigmas; /* extension (tests): LJ cutoff in sigmas; 5 = the reference (ljForce.c:114), 2.5 = upstream CoMD, whose cohesive energy CoMD.c:897 documents */
} Command;

Command parseCommandLine(int argc, char** argv);
void printCmdYaml(FILE* file, Command* cmd);

/* ---- parallel layer: parallel.h.  One process per GPU; the transport is pluggable. ---- */
/* CommTransport: include/comd_hip.h */

void initParallel(int rank, int nRanks, const CommTransport* transport);   /* parallel.c:57-64 */
void destroyParallel(void);
int  loopbackParallel(void);
int  getNRanks(void);
int  getMyRank(void);
int  printRank(void);
void barrierParallel(void);
void timestampBarrier(const char* msg);
int  sendReceiveParallel(void* sendBuf, int sendLen, int dest, void* recvBuf, int recvLen, int source);
int  sendReceiveDevice(void* sendBuf, int sendLen, int dest, void* recvBuf, int recvLen, int source, comdStream_t stream);
void sendReceiveDevice2(void* sendM, int nSendM, int dstM, void* recvP, void* sendP, int nSendP, int dstP, void* recvM,
                        int recvCap, comdStream_t stream, int nRecv[2]);
int  sizedExchangeAvailable(void);
void sendReceiveDevice2Sized(void* sendM, int nSendM, int dstM, void* recvP, int nRecvP, void* sendP, int nSendP, int dstP, void* recvM, int nRecvM,
                             comdStream_t stream);
void addIntParallel(int* sendBuf, int* recvBuf, int count);
void addRealParallel(real_t* sendBuf, real_t* recvBuf, int count);
void addDoubleParallel(double* sendBuf, double* recvBuf, int count);
void maxIntParallel(int* sendBuf, int* recvBuf, int count);
void bcastParallel(void* buf, int len, int root);

/* ---- decomposition.h ---- */
typedef struct DomainSt {
   int procGrid[3], procCoord[3];
   real_t globalMin[3], globalMax[3], globalExtent[3];
   real_t localMin[3], localMax[3], localExtent[3];
} Domain;
Domain* initDecomposition(int xproc, int yproc, int zproc, const real_t globalExtent[3]);
int processorNum(Domain* domain, int dix, int diy, int diz);

/* ---- linkCells.h ---- */
typedef struct LinkCellSt {
   int gridSize[3];
   int nLocalBoxes, nHaloBoxes, nTotalBoxes;
   real_t localMin[3], localMax[3], boxSize[3], invBoxSize[3];
   int* nAtoms;           /* host copy; refreshed from the device by sumAtoms */
   int maxAtoms;          /* slot capacity per cell */
   int *boxIDLookUp, *boxIDLookUpReverse;   /* -H: Hilbert numbering of the local cells (linkCells.h:27-28); NULL otherwise */
   CellGeom geom;
} LinkCell;
LinkCell* initLinkCells(const Domain* domain, real_t cutoff, int useHilbert);
void destroyLinkCells(LinkCell** boxes);
int getNeighborBoxes(LinkCell* boxes, int iBox, int* nbrBoxes);
int getBoxFromTuple(LinkCell* boxes, int x, int y, int z);
int getBoxFromCoord(LinkCell* boxes, const real_t rr[3]);
int maxOccupancy(LinkCell* boxes);

/* ---- initAtoms.h ---- */
typedef struct AtomsSt {
   int nLocal, nGlobal;
   HostAtoms h;           /* slot arrays, cell*maxAtoms + i */
} Atoms;
struct SimFlatSt;
Atoms* initAtoms(LinkCell* boxes);
void destroyAtoms(Atoms* atoms);
/* pass 1 (atoms == NULL storage): returns the largest per-cell count the lattice produces on this rank */
int  countFccLattice(int nx, int ny, int nz, real_t lat, const Domain* domain, LinkCell* boxes);
void createFccLattice(int nx, int ny, int nz, real_t lat, struct SimFlatSt* s);
void setTemperature(struct SimFlatSt* s, real_t temperature);
void randomDisplacements(struct SimFlatSt* s, real_t delta);
int  putAtomInBox(LinkCell* boxes, Atoms* atoms, int gid, int iType, real_t x, real_t y, real_t z, real_t px, real_t py, real_t pz);
void kineticEnergyHost(struct SimFlatSt* s);   /* timestep.c:109-133, used during initialisation only */

/* ---- random.h ---- */
real_t   gasdev(uint64_t* seed);
double   lcg61(uint64_t* seed);
uint64_t mkSeed(uint32_t id, uint32_t callSite);

/* ---- potentials: CoMDTypes.h:42-53, 140-201 ---- */
typedef struct BasePotentialSt {
   real_t cutoff, mass, lat;
   char latticeType[8];
   char name[3];
   int atomicNo;
   int  (*force)(struct SimFlatSt* s);
   void (*print)(FILE* file, struct BasePotentialSt* pot);
   void (*destroy)(struct BasePotentialSt** pot);
} BasePotential;

typedef struct LjPotentialSt {
   real_t cutoff, mass, lat; char latticeType[8]; char name[3]; int atomicNo;
   int  (*force)(struct SimFlatSt* s);
   void (*print)(FILE* file, BasePotential* pot);
   void (*destroy)(BasePotential** pot);
   real_t sigma, epsilon;
} LjPotential;

typedef struct InterpolationObjectSt {
   int n; real_t x0, invDx; real_t* values; real_t invDxXx0;
} InterpolationObject;

struct HaloExchangeSt;
typedef struct EamPotentialSt {
   real_t cutoff, mass, lat; char latticeType[8]; char name[3]; int atomicNo;
   int  (*force)(struct SimFlatSt* s);
   void (*print)(FILE* file, BasePotential* pot);
   void (*destroy)(BasePotential** pot);
   InterpolationObject *phi, *rho, *f;
   real_t *phiSpline, *rhoSpline;     /* -P: 4 * n cubic-spline coefficients in r^2 each (gpu_utility.c:377-430), else NULL */
   struct HaloExchangeSt* forceExchange;
} EamPotential;

BasePotential* initLjPot(void);
BasePotential* initEamPot(const char* dir, const char* file, const char* type);
void interpolate(InterpolationObject* table, real_t r, real_t* f, real_t* df);
real_t* initSplineCoefficients(int n, const real_t* values, real_t x0, real_t invDx);   /* malloc'd, 4 * n */
void eamUseSplines(BasePotential* pot);                                                /* fill phiSpline / rhoSpline */

typedef struct SpeciesDataSt { char name[3]; int atomicNo; real_t mass; } SpeciesData;
typedef struct ValidateSt { double eTot0; int nAtoms0; } Validate;

/* ---- halo exchange: haloExchange.h ---- */
enum HaloFaceOrder { HALO_X_MINUS, HALO_X_PLUS, HALO_Y_MINUS, HALO_Y_PLUS, HALO_Z_MINUS, HALO_Z_PLUS };
enum HaloAxisOrder { HALO_X_AXIS, HALO_Y_AXIS, HALO_Z_AXIS };

/* Per axis: the counts (atoms) of the four messages of the previous exchange -- sent through the minus face, sent through the plus face,
 * received from the plus neighbour, received from the minus neighbour -- mirrored by a one-wave kernel into pinned memory.  Sender and
 * receiver of a message hold the same number, so both size the next transfer as count + 12.5 % + 64 without talking to each other. */
typedef struct HaloSpecSt {
   int   valid;                           /* mirror[] describes the previous exchange of this axis */
   int   pending;                         /* the mirror kernel has been enqueued: wait for `event` before reading */
   int*  mirror;                          /* pinned [4] */
   void* event;
   int   haveBound, lastBound[4];         /* sizes the previous exchange was posted with: a count above its bound means that exchange was cut short */
   int   lastCount[4];                    /* ... and the counts those sizes were derived from (both ends of a message hold the same two numbers) */
} HaloSpec;

typedef struct HaloExchangeSt {
   int nbrRank[6];
   int bufCapacity;                       /* bytes per message buffer */
   int (*loadBuffer)(void* parms, void* data, int face, char* buf);       /* returns bytes, or -1: "ask msgBytes when you need it" */
   int (*msgBytes)(void* parms, void* data, int face, char* buf);         /* optional: size of a packed device message (blocks) */
   /* optional: both messages of an axis phase behind one synchronisation */
   void (*msgBytes2)(void* parms, void* data, int faceM, char* bufM, int faceP, char* bufP, int out[2]);
   void (*unloadBuffer)(void* parms, void* data, int face, int bufSize, char* buf);
   /* optional: both faces of an axis phase packed / unpacked by one device launch each (same results as two loadBuffer / unloadBuffer calls) */
   void (*loadBuffer2)(void* parms, void* data, int faceM, char* bufM, int faceP, char* bufP, int nBytes[2]);
   void (*unloadBuffer2)(void* parms, void* data, int faceA, int bufSizeA, char* bufA, int faceB, int bufSizeB, char* bufB);
   void (*destroy)(void* parms);
   void* parms;
   int type;                              /* 0 atoms, 1 force, 2 positions */
   int deviceBuffers;                     /* 1: the four buffers are device memory */
   char *sendBufM, *sendBufP, *recvBufM, *recvBufP;
   /* size agreement without a handshake (device buffers, messages that leave the rank) */
   int msgHeaderBytes, msgBytesPerAtom;   /* bytes of a message of n atoms = msgHeaderBytes + n * msgBytesPerAtom */
   int capacityAtoms;
   int exactCounts;                       /* counts cannot change between invalidations (positions between list builds): no slack */
   /* optional: device addresses of the four counts of axis phase (faceM, faceP), in HaloSpec order */
   void (*countPtrs)(void* parms, struct HaloExchangeSt* hh, int faceM, int faceP, const int* out[4]);
   /* optional: tell the plugin the agreed sizes (atoms) of the message it packs for `face` and of the one it unpacks for `face`; 0 = none */
   void (*setBounds)(void* parms, int face, int sendBoundAtoms, int recvBoundAtoms);
   HaloSpec spec[3];
   /* optional: fill the halo cells of axes firstAxis..2 (all of them self-neighbour axes) straight from their sources, one launch (comd_hip.h mirrorSlotCellsGpu) */
   void (*mirror)(struct HaloExchangeSt* hh, void* data, int firstAxis);
   int nTotalBoxes;                       /* (for the mirror map) */
} HaloExchange;

typedef struct AtomExchangeParmsSt {
   int nCells[6];
   int* cellList[6];                      /* host */
   int* cellListGpu[6];                   /* device */
   int* d_cellOffsets;                    /* device scratch, max nCells + 1 */
   int* d_cellOffsets2;                   /* second scratch: the two faces of an axis phase are scanned and packed together */
   real_t shift[6][3];                    /* pbcFactor * globalExtent */
   int capacityAtoms;
   int sendBound[6], recvBound[6];        /* agreed message sizes in atoms, 0 = none (capacityAtoms applies) */
} AtomExchangeParms;

typedef struct ForceExchangeParmsSt {
   int nCells[6];
   int *sendCells[6], *recvCells[6];
   int *sendCellsGpu[6], *recvCellsGpu[6];
   int *sendOffsetsGpu[6], *recvOffsetsGpu[6];   /* device, nCells + 1 each: filled by one batched scan per step */
   int* d_cellOffsets;
   int capacityAtoms;
   int positions;                         /* 0: dF/drho (1 real per atom); 1: positions + face shift (3 reals per atom) */
   /* mirror map of the self-neighbour tail firstAxis..2 (built on first use): halo cell mirrorDst[k] is the image of mirrorSrc[k] displaced by mirrorShift[3k..] */
   int mirrorFirst, mirrorPairs;
   int *mirrorDstGpu, *mirrorSrcGpu; real_t* mirrorShiftGpu;
   real_t shift[6][3];
   int msgBytesCached[6];                 /* positions: message sizes are fixed between list builds; -1 = unknown */
   int sendBound[6];                      /* agreed message sizes in atoms, 0 = none */
} ForceExchangeParms;

HaloExchange* initAtomHaloExchange(Domain* domain, LinkCell* boxes, int allocDevice);
HaloExchange* initForceHaloExchange(Domain* domain, LinkCell* boxes, int allocDevice);
/* Verlet-list mode: refresh of the halo copies' positions between list builds, over the force exchange's cell lists */
HaloExchange* initPositionHaloExchange(Domain* domain, LinkCell* boxes, int allocDevice);
void preparePositionExchange(HaloExchange* positionExchange, struct SimFlatSt* sim);   /* after every list build */
void destroyHaloExchange(HaloExchange** haloExchange);
void invalidateHaloSizes(HaloExchange* haloExchange);      /* the next exchange of every axis swaps exact sizes again */
void haloExchange(HaloExchange* haloExchange, void* data);
void comdSetHaloHandshake(int on);                          /* 1: swap exact message sizes before every exchange (COMD_HALO_HANDSHAKE=1), 0: sized protocol, -1: ask the environment */
int  haloMirrorFirstAxis(const HaloExchange* hh);         /* first axis of the self-neighbour tail the plugin mirrors directly; 3 = none */
void exchangeData(HaloExchange* haloExchange, void* data, int iAxis);
void prepareForceExchange(HaloExchange* forceExchange, struct SimFlatSt* sim);   /* one batched scan of all twelve cell lists */
int* mkAtomCellList(LinkCell* boxes, enum HaloFaceOrder iFace, int nCells);
int* mkForceSendCellList(LinkCell* boxes, int face, int nCells);
int* mkForceRecvCellList(LinkCell* boxes, int face, int nCells);

/* ---- simulation: CoMDTypes.h:75-135 ---- */
typedef struct SimFlatSt {
   int nSteps, printRate;
   double dt;
   Domain* domain;
   LinkCell* boxes;
   Atoms* atoms;
   SpeciesData* species;
   real_t ePotential, eKinetic;
   BasePotential* pot;
   HaloExchange* atomExchange;
   HaloExchange* positionExchange;  /* *_nl methods only */
   SimGpu gpu;
   int method;
   int n_boundary_cells, n_boundary1_cells;
   int *boundary_cells_h, *interior_cells_h, *boundary1_cells_h;
   int gpuAsync, gpuProfile;
   int ljInterpolation, spline, usePairlist;
   real_t skinDistance;
   int useNL;                       /* method is thread_atom_nl / warp_atom_nl */
   int interiorLaunched;            /* -a 1: redistributeAtoms has already started the interior cells' force work */
   int nlBuilds;                    /* list builds so far (reported) */
   int quiet, cmdDoeam;
   int iStepPrev, firstPrint;       /* printThings state (static locals in CoMD.c:466-467) */
} SimFlat;

SimFlat* initSimulation(Command cmd);
SimFlat* initSimulationHost(Command cmd);      /* no device: lattice, momenta, link cells only */
void destroySimulation(SimFlat** ps);
void sumAtoms(SimFlat* s);
void printThings(SimFlat* s, int iStep, double elapsedTime);
Validate* initValidate(SimFlat* s);
void validateResult(const Validate* val, SimFlat* sim);
void printSimulationDataYaml(FILE* file, SimFlat* s);
void setBoundaryCellsHost(SimFlat* sim, HaloExchange* hh);

/* ---- timestep.h ---- */
double timestep(SimFlat* s, int n, real_t dt);
void computeForce(SimFlat* s);
void kineticEnergyGpu(SimFlat* s);
void redistributeAtoms(SimFlat* sim);
void ensureInteriorForceLaunched(SimFlat* sim);

/* ---- performanceTimers.h ---- */
enum TimerHandle { totalTimer, loopTimer, timestepTimer, positionTimer, velocityTimer, redistributeTimer, atomHaloTimer,
                   computeForceTimer, eamHaloTimer, commHaloTimer, commReduceTimer, neighborListBuildTimer, numberOfTimers };
void profileStart(enum TimerHandle handle);
void profileStop(enum TimerHandle handle);
double getElapsedTime(enum TimerHandle handle);
void resetTimers(void);
/* device-true timing: every start/stop of a phase of timestep() also records a HIP event on `stream`; the report then shows device seconds for those
 * phases (same table, same YAML keys) and adds atomUpdatesPerSec / forceKernelGBs.  forceBytesPerAtom: algorithmic bytes of one force evaluation. */
void timersUseDevice(int on, comdStream_t stream, double forceBytesPerAtom);
void printPerformanceResults(int nGlobalAtoms, int printRate);
void printPerformanceResultsYaml(FILE* file);
#define startTimer(h) profileStart(h)
#define stopTimer(h)  profileStop(h)

/* ---- yamlOutput.h ---- */
extern FILE* yamlFile;
void yamlBegin(void);
void yamlEnd(void);
void yamlAppInfo(FILE* file);
void printSeparator(FILE* file);

/* ---- embedding API (Python/ctypes, tests, bench): a thin layer over the functions above ---- */
SimFlat* comdCreate(int argc, char** argv);              /* parse the reference CLI, initSimulation */
SimFlat* comdCreateHostOnly(int argc, char** argv);      /* parse + initSimulationHost (CPU tests) */
const HostAtoms* comdHostAtoms(SimFlat* s);              /* the host mirror, without touching the device */
int      comdSimBoxFromTuple(SimFlat* s, int ix, int iy, int iz);
int      comdSimBoxFromCoord(SimFlat* s, const double r[3]);
int      comdFaceCells(SimFlat* s, int kind, int face, int* list);
void     comdNeighborRanks(SimFlat* s, int nbr[6], int coord[3]);
void     comdHaloExchangeHost(SimFlat* s, int (*load)(void*, void*, int, char*), void (*unload)(void*, void*, int, int, char*));
void     comdFaceShift(SimFlat* s, int face, double out[3]);
int      comdPutAtomInBox(SimFlat* s, int gid, int type, const double r[3], const double p[3]);
int      comdEamTable(SimFlat* s, int which, double* x0, double* invDx, double* values);   /* 0 phi, 1 rho, 2 F; n + 3 padded samples */
int      comdNeighborListBuilds(SimFlat* s);              /* Verlet-list builds so far (*_nl methods) */
void     comdGridInfo(SimFlat* s, int out[6]);            /* gridSize[3], nLocalBoxes, nTotalBoxes, maxAtoms */
int      comdMain(int argc, char** argv);                /* the reference's main(): CoMD.c:86-187 */
void     comdDestroy(SimFlat* s);
SimGpu*  comdSimGpu(SimFlat* s);                          /* the device half, for the comd_hip.h calls that take a SimGpu* (force timing) */
void     comdGetEnergy(SimFlat* s, double out[3]);       /* ePotential, eKinetic, nGlobal */
int      comdNumGlobal(SimFlat* s);
int      comdNumLocalSlots(SimFlat* s);                  /* nTotalBoxes * maxAtoms */
/* copy device state to the host mirror and return pointers into it (valid until the next call) */
const HostAtoms* comdFetchAtoms(SimFlat* s);
/* per-atom arrays of this rank's local atoms keyed by gid: which 0 r,1 p,2 f (3 doubles) / 3 e, 4 rhobar, 5 dfEmbed (1 double).
 * `out` must be zero-initialised by the caller when ranks are to be summed. */
void     comdGatherByGid(SimFlat* s, int which, double* out);
/* overwrite r or p of local atoms from a by-gid array (which 0 / 1), then upload */
void     comdScatterByGid(SimFlat* s, int which, const double* in);

#endif
