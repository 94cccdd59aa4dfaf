/* yaml_report.c -- the YAML side file and the application-info block (yamlOutput.c:45-128,
 * generate_info_header:59-74).  Field names and layout follow the reference; platform/build strings are
 * gathered at run time (uname) instead of a generated header. */
#include "comd_host.h"
#include <time.h>
#include <string.h>
#include <sys/utsname.h>

FILE* yamlFile = NULL;
static const char* CoMDVersion = "1.1";
static const char* CoMDVariant = "CoMD-hip-gfx950";

#ifndef COMD_BUILD_CC
#define COMD_BUILD_CC "gcc + hipcc"
#endif
#ifndef COMD_BUILD_CFLAGS
#define COMD_BUILD_CFLAGS "-std=gnu11 -O2 / -O3 --offload-arch=gfx950"
#endif

static void getTimeString(char* out)
{
   time_t raw; time(&raw);
   struct tm* t = localtime(&raw);
   sprintf(out, "%04d-%02d-%02d, %02d:%02d:%02d", t->tm_year + 1900, t->tm_mon + 1, t->tm_mday, t->tm_hour, t->tm_min, t->tm_sec);
}

void yamlBegin(void)
{
   if (!printRank()) return;
   char filename[160], sdate[96];
   time_t raw; time(&raw);
   struct tm* t = localtime(&raw);
   sprintf(sdate, "%04d:%02d:%02d-%02d:%02d:%02d", t->tm_year + 1900, t->tm_mon + 1, t->tm_mday, t->tm_hour, t->tm_min, t->tm_sec);
   sprintf(filename, "%s.%s.yaml", CoMDVariant, sdate);
   yamlFile = fopen(filename, "w");
}

void yamlEnd(void)
{
   if (!printRank() || !yamlFile) return;
   fclose(yamlFile);
   yamlFile = NULL;
}

void printSeparator(FILE* file) { if (file) fprintf(file, "\n"); }

void yamlAppInfo(FILE* file)
{
   if (!printRank() || !file) return;
   struct utsname u; memset(&u, 0, sizeof u); uname(&u);
   printSeparator(file);
   fprintf(file, "Mini-Application Name    : %s\n", CoMDVariant);
   fprintf(file, "Mini-Application Version : %s\n", CoMDVersion);
   fprintf(file, "Platform:\n");
   fprintf(file, "  hostname: %s\n", u.nodename);
   fprintf(file, "  kernel name: '%s'\n", u.sysname);
   fprintf(file, "  kernel release: '%s'\n", u.release);
   fprintf(file, "  processor: '%s'\n", u.machine);
   fprintf(file, "Build:\n");
   fprintf(file, "  CC: '%s'\n", COMD_BUILD_CC);
   fprintf(file, "  compiler version: '%s'\n", __VERSION__);
   fprintf(file, "  CFLAGS: '%s'\n", COMD_BUILD_CFLAGS);
   fprintf(file, "  LDFLAGS: '-lcomd_hip -lm'\n");
   fprintf(file, "  using MPI: false\n");
   fprintf(file, "  Threading: none\n");
   fprintf(file, "  Double Precision: %s\n", (sizeof(real_t) == sizeof(double) ? "true" : "false"));
   char ts[96]; getTimeString(ts);
   fprintf(file, "Run Date/Time: %s\n", ts);
   fprintf(file, "\n");
   fflush(file);
}
