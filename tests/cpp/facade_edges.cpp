// The edges of Decoder::setupDecimationStagesFactor / ...BW (reference code/Decoder/Decoder.h:268-412) through the facade, without a GPU: no engine
// exists before the first process(), so this runs anywhere the library loads.  tests/test_facade_edges.py holds the expected transcript.
#include "habdec/Decoder.h"
#include <cstdio>
int main(){
  habdec::Decoder<float> d;
  habdec::IQVector<float> v; v.resize(16); v.samplingRate(2048000.0);
  d.pushSamples(v);
  size_t a = d.setupDecimationStagesFactor(64);  printf("R %zu %d\n", a, d.getDecimationFactor());
  a = d.setupDecimationStagesFactor(3);   printf("R %zu %d\n", a, d.getDecimationFactor());
  a = d.setupDecimationStagesFactor(16);  printf("R %zu %d\n", a, d.getDecimationFactor());
  a = d.setupDecimationStagesFactor(1);   printf("R %zu %d\n", a, d.getDecimationFactor());
  a = d.setupDecimationStagesFactor(8);   printf("R %zu %d\n", a, d.getDecimationFactor());
  a = d.setupDecimationStagesFactor(0);   printf("R %zu %d\n", a, d.getDecimationFactor());
  a = d.setupDecimationStagesFactor(512); printf("R %zu %d\n", a, d.getDecimationFactor());
  a = d.setupDecimationStagesBW(40000.0); printf("R %zu %d\n", a, d.getDecimationFactor());
  a = d.setupDecimationStagesBW(4e6);     printf("R %zu %d\n", a, d.getDecimationFactor());
  a = d.setupDecimationStagesBW(100.0);   printf("R %zu %d\n", a, d.getDecimationFactor());
}
