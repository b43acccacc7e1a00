#include <hip/hip_runtime_api.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
static double now_ms(void){struct timespec t;clock_gettime(CLOCK_MONOTONIC,&t);return t.tv_sec*1e3+t.tv_nsec*1e-6;}
int main(){
  size_t n=28000000; void*d; hipMalloc(&d,n); hipStream_t s; hipStreamCreate(&s);
  for(int r=0;r<6;++r){
    char*h=malloc(n); memset(h,1,n);
    double t0=now_ms(); hipError_t e=hipHostRegister(h,n,hipHostRegisterDefault); double t1=now_ms();
    hipMemcpyAsync(d,h,n,hipMemcpyHostToDevice,s); hipStreamSynchronize(s); double t2=now_ms();
    hipHostUnregister(h); double t3=now_ms();
    hipMemcpy(d,h,n,hipMemcpyHostToDevice); double t4=now_ms();
    printf("register %.3f ms (%d), pinned H2D %.3f ms, unregister %.3f ms, pageable H2D %.3f ms\n",t1-t0,(int)e,t2-t1,t3-t2,t4-t3);
    free(h);
  }
  return 0;}
