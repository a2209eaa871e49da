#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
static inline float asf(uint32_t u){float f;memcpy(&f,&u,4);return f;}
static inline uint32_t asu(float f){uint32_t u;memcpy(&u,&f,4);return u;}
static float my_acosf(float x){
  const float one=1.0f, pi=3.1415925026e+00f, pio2_hi=1.5707962513e+00f, pio2_lo=7.5497894159e-08f,
  pS0=1.6666667163e-01f,pS1=-3.2556581497e-01f,pS2=2.0121252537e-01f,pS3=-4.0055535734e-02f,pS4=7.9153501429e-04f,pS5=3.4793309169e-05f,
  qS1=-2.4033949375e+00f,qS2=2.0209457874e+00f,qS3=-6.8828397989e-01f,qS4=7.7038154006e-02f;
  float z,p,q,r,w,s,c,df; int32_t hx=(int32_t)asu(x), ix=hx&0x7fffffff;
  if(ix==0x3f800000){ if(hx>0) return 0.0f; else return pi+2.0f*pio2_lo; }
  else if(ix>0x3f800000) return (x-x)/(x-x);
  if(ix<0x3f000000){ if(ix<=0x23000000) return pio2_hi+pio2_lo;
    z=x*x; p=z*(pS0+z*(pS1+z*(pS2+z*(pS3+z*(pS4+z*pS5))))); q=one+z*(qS1+z*(qS2+z*(qS3+z*qS4))); r=p/q; return pio2_hi-(x-(pio2_lo-x*r)); }
  else if(hx<0){ z=(one+x)*0.5f; p=z*(pS0+z*(pS1+z*(pS2+z*(pS3+z*(pS4+z*pS5))))); q=one+z*(qS1+z*(qS2+z*(qS3+z*qS4))); s=sqrtf(z); r=p/q; w=r*s-pio2_lo; return pi-2.0f*(s+w); }
  else { z=(one-x)*0.5f; s=sqrtf(z); df=asf(asu(s)&0xfffff000u); c=(z-df*df)/(s+df); p=z*(pS0+z*(pS1+z*(pS2+z*(pS3+z*(pS4+z*pS5))))); q=one+z*(qS1+z*(qS2+z*(qS3+z*qS4))); r=p/q; w=r*s+c; return 2.0f*(df+w); }
}
int main(){
  long bad=0,n=0; 
  for(uint32_t sgn=0; sgn<2; sgn++)
    for(uint32_t u=0; u<=0x3f800000u; u+=3){ float x=asf(u|(sgn<<31)); volatile float a=acosf(x); float b=my_acosf(x); n++; if(asu(a)!=asu(b)){ if(bad<5) printf("x=%a libm=%a mine=%a\n",x,a,b); bad++; } }
  printf("%ld of %ld differ\n",bad,n); return 0; }
