// Full sweep behind mrca_device.h:norm_obs: every float in [0, 6].  gcc -O2 -mfma -ffp-contract=off tools/check_div6.c -lm
#include <stdio.h>
#include <stdint.h>
#include <string.h>
#include <math.h>
int main(){
  const float inv6 = 1.0f/6.0f; // correctly rounded
  uint64_t bad=0, bad2=0; uint32_t first=0,last=0;
  for(uint32_t u=0; u<=0x40C00000u; ++u){
    float x; memcpy(&x,&u,4);
    float want = x/6.0f;
    float q = x*inv6;
    float r = __builtin_fmaf(-q,6.0f,x);
    float q2 = __builtin_fmaf(r,inv6,q);
    if(memcmp(&q2,&want,4)){ if(!bad) first=u; bad++; if(u>last) last=u; }
    // normalised obs
    float a = want-0.5f, b=q2-0.5f; if(memcmp(&a,&b,4)) bad2++;
  }
  printf("mismatches %llu first %08x last %08x obs mismatches %llu\n",(unsigned long long)bad, first,last,(unsigned long long)bad2);
  return 0;
}
