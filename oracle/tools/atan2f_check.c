/*
 * atan2f_check.c -- TEST INFRASTRUCTURE (oracle/): the restated float atan2 of the LiDAR-Iris image (oracle/iris_oracle.c: iriso_atan2f =
 * glibc's e_atan2f.c around the pinned atanf; the device's copy: csrc/iris.hip) against THIS platform's libm atan2f on 4e9 pairs:
 * random bit patterns, every pair out of sixteen special values (zeros, denormals, 1, infinities, NaN, extremes), one special
 * against random, and coordinates as a scan has them (+-100 m).  Prints the number of differing results; exit code 1 if any.
 *   make -C oracle atan2f-check
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <pthread.h>
#include "iris_oracle.h"
static inline uint32_t f2u(float f){uint32_t u;memcpy(&u,&f,4);return u;}
static inline float u2f(uint32_t u){float f;memcpy(&f,&u,4);return f;}
typedef struct { uint64_t seed; uint64_t n; uint64_t bad; uint32_t by, bx; } job_t;
static void *run(void *a){ job_t *j=a; uint64_t s=j->seed;
  static const uint32_t special[]={0,0x80000000u,1,0x80000001u,0x007fffff,0x00800000,0x3f800000,0xbf800000,0x7f800000,0xff800000,0x7fc00000,0x7f7fffff,0xff7fffff,0x3f000000,0x40490fdb,0x1e3ce508};
  for(uint64_t i=0;i<j->n;i++){ s=s*6364136223846793005ull+1442695040888963407ull; uint64_t r=s^(s>>29);
    uint32_t by=(uint32_t)(r>>32), bx=(uint32_t)r;
    int mode=i&7;
    if(mode==0){ by=special[(r>>8)&15]; } else if(mode==1){ bx=special[(r>>12)&15]; } else if(mode==2){ by=special[(r>>8)&15]; bx=special[(r>>12)&15]; }
    else if(mode>=5){ /* realistic coordinates */ float fy=(float)((double)(int32_t)by/2147483648.0*100.0), fx=(float)((double)(int32_t)bx/2147483648.0*100.0); by=f2u(fy); bx=f2u(fx); }
    float y=u2f(by), x=u2f(bx); float a=iriso_atan2f(y,x), b=atan2f(y,x);
    uint32_t ua=f2u(a), ub=f2u(b); if(a!=a) ua=0x7fc00000; if(b!=b) ub=0x7fc00000;
    if(ua!=ub){ if(!j->bad){j->by=by;j->bx=bx;} j->bad++; } }
  return 0; }
int main(){ pthread_t th[8]; job_t jobs[8]; for(int t=0;t<8;t++){ jobs[t]=(job_t){12345+t*7919ull, 500000000ull,0,0,0}; pthread_create(&th[t],0,run,&jobs[t]); }
  uint64_t bad=0; for(int t=0;t<8;t++){ pthread_join(th[t],0); bad+=jobs[t].bad; if(jobs[t].bad) printf("first diff y=%08x x=%08x mine %.9g libm %.9g\n", jobs[t].by, jobs[t].bx, iriso_atan2f(u2f(jobs[t].by),u2f(jobs[t].bx)), atan2f(u2f(jobs[t].by),u2f(jobs[t].bx))); }
  printf("4e9 pairs, differences: %llu\n",(unsigned long long)bad); return bad!=0; }
