// Sanitizer harness for the gzip decoders of the ingest path (tests/test_host_cpu.py builds it with -fsanitize=thread
// and with -fsanitize=address,undefined): several decodes with different thread counts and chunkings, plus a decoder that
// is abandoned in the middle of the stream.
#include "../../genestrip_amd/csrc/gs_inflate.h"
#include <cstdio>
#include <vector>
#include <random>
#include <string>
static std::vector<uint8_t> gz(const std::vector<uint8_t>&in,int level){ z_stream z{}; deflateInit2(&z,level,Z_DEFLATED,31,8,Z_DEFAULT_STRATEGY); std::vector<uint8_t> out(deflateBound(&z,in.size())+64); z.next_in=(Bytef*)in.data(); z.avail_in=in.size(); z.next_out=out.data(); z.avail_out=out.size(); deflate(&z,Z_FINISH); out.resize(z.total_out); deflateEnd(&z); return out;}
static std::vector<uint8_t> fastq(size_t n,uint64_t seed){ std::mt19937_64 rng(seed); std::vector<uint8_t> v; const char*b="ACGT"; unsigned long long id=0;
  while(v.size()<n){char d[96]; int m=snprintf(d,96,"@A00123:45:HXX:1:1101:%llu:%llu 1:N:0:ACGT\n",1000+(id/50)%30000,1000+(id*37)%40000); id++; v.insert(v.end(),d,d+m); for(int i=0;i<150;i++)v.push_back(b[rng()&3]); v.push_back('\n');v.push_back('+');v.push_back('\n'); for(int i=0;i<150;i++){unsigned r=rng()%100; v.push_back(r<88?'F':':');} v.push_back('\n');} v.resize(n); return v;}
// BGZF: members of <= 64 KiB of text with a 'BC' extra subfield holding the member size - 1
static std::vector<uint8_t> bgzf(const std::vector<uint8_t>&in){ std::vector<uint8_t> out; const size_t B=65280;
  for(size_t at=0;;at+=B){ const size_t n=at<in.size()?std::min(B,in.size()-at):0; z_stream z{}; deflateInit2(&z,6,Z_DEFLATED,-15,8,Z_DEFAULT_STRATEGY);
    std::vector<uint8_t> body(deflateBound(&z,n)+16); z.next_in=(Bytef*)in.data()+(n?at:0); z.avail_in=n; z.next_out=body.data(); z.avail_out=body.size(); deflate(&z,Z_FINISH); body.resize(z.total_out); deflateEnd(&z);
    const uint32_t bsize=(uint32_t)(12+6+body.size()+8-1), crc=(uint32_t)crc32(crc32(0,Z_NULL,0),in.data()+(n?at:0),n), isz=(uint32_t)n;
    const uint8_t h[18]={0x1f,0x8b,8,4,0,0,0,0,0,0xff,6,0,'B','C',2,0,(uint8_t)bsize,(uint8_t)(bsize>>8)}; out.insert(out.end(),h,h+18); out.insert(out.end(),body.begin(),body.end());
    for(int i=0;i<4;i++)out.push_back((uint8_t)(crc>>(8*i))); for(int i=0;i<4;i++)out.push_back((uint8_t)(isz>>(8*i)));
    if(n==0)break; }
  return out;}
int main(){ auto in=fastq(6000000,1); auto c=gz(in,6); int fails=0;
  { auto b=bgzf(in); for(int threads:{1,3,8}) for(size_t room:{200000ul,65536ul,70001ul}){ GsBgzfReader br(b.data(),b.size(),threads); std::vector<uint8_t> got,buf(room); bool done=false; while(!done){size_t p=0; if(!br.read(buf.data(),buf.size(),&p,&done)){fails++;break;} got.insert(got.end(),buf.begin(),buf.begin()+p);} if(got!=in||br.rest_offset()!=b.size())fails++; } }
  for(int rep=0;rep<3;rep++) for(int threads:{2,5}) for(size_t chunk:{65536ul,300000ul}){ GsParallelGunzip pg; pg.start(c.data(),c.size(),threads,chunk); std::vector<uint8_t> got,buf(200000); bool done=false; while(!done){size_t p=0; if(!pg.read(buf.data(),buf.size(),&p,nullptr,&done)){fails++;break;} got.insert(got.end(),buf.begin(),buf.begin()+p);} if(got!=in)fails++; }
  { GsParallelGunzip pg; pg.start(c.data(),c.size(),4,65536); std::vector<uint8_t> buf(100000); size_t p; bool done; pg.read(buf.data(),buf.size(),&p,nullptr,&done); /* abandon mid-stream: destructor must stop cleanly */ }
  printf("fails %d\n",fails); return fails; }
