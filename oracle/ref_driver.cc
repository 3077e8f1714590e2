// oracle/ref_driver.cc -- TEST INFRASTRUCTURE, not product code.
//
// A small driver of OUR OWN that is compiled against the reference's headers and
// .cc files *where they lie* under /root/reference/lib/assembly/src (see
// oracle/build_ref.sh).  Nothing from the reference is copied into this repo.
//
// Why a driver and not the reference's own createDict():  the translation unit
// paths/long/BuildReadQGraph48.cc cannot be compiled unmodified by either
// compiler in this image (g++ 11: graph/Digraph.h:1450,1461 and
// kmers/KmerShape.h:566 are ill-formed templates; clang 22: paths/KmerPathInterval.h
// friend default arguments), and its hot-path functions are file-static.  What DOES
// compile in place is every component those ~250 lines of glue are built from:
//   KMer<K> (kmers/KMer.h), KMerContext (kmers/KMerContext.{h,cc}),
//   CF<K>::getForm (dna/CanonicalForm.h), FNV1a (math/Hash.h),
//   KDef / KmerDictEntry / KmerDict::recomputeAdjacencies / KmerVec (kmers/ReadPather.h),
//   HashSet (feudal/HashSet.h), MapReduceEngine (MapReduceEngine.h),
//   PQVecEncoder/PQVec (feudal/PQVec.{h,cc}), BaseVec / MasterVec / feudal file IO,
//   BinaryWriter/BinaryReader (feudal/BinaryStream.h).
// This driver therefore calls the REAL reference classes for all arithmetic, data
// layout, hashing, sorting/grouping, hash-set lookup and file IO, and restates only
// the glue (BuildReadQGraph48.cc:63-80 tail finder, :134-190 Kmerizer, :211-318
// createDict flow) in its own words.  DESIGN.md calls this pin "reference components,
// restated glue".
//
// Sub-commands (all paths are files; see tests/golden/make_golden.py for usage):
//   kat                                  known-answer prints for KMer<40|48|60>
//   mkreads  in.raw out_head             raw reads/quals -> out_head.{fastb,qualp} via
//                                        the reference's BaseVec/PQVecEncoder/feudal writer
//   rdreads  head out.raw                reference reader -> raw reads/quals
//   side     head outdir                 .lens/.qhist/.dti/subsam.* as DF's ingest writes them (reference BinaryWriter)
//   graph ... (same arguments as dict)   dict, then the unipath edges (edges.fastb) and outdir/a.<K>/{a.fastb,a.hbv,a.hbx,
//                                        a.kmers,a.inv,a.to_left,a.to_right,a.k} -- see oracle/ref_graph.cc
//   dict K head outdir minQual minFreq minBC useBC nThreads [ignBcBelow]
//                                        goodlens.u32, kmers.kvec (pre-adjacency, written by
//                                        the reference's BinaryWriter), solid.bin (post
//                                        recomputeAdjacencies, sorted), spectrum.txt, times.txt
//
// raw format: u64 nReads, then per read: u32 len, len base codes (0..3), len quals.

#include "MapReduceEngine.h"
#include "Basevector.h"
#include "Qualvector.h"
#include "feudal/PQVec.h"
#include "feudal/BinaryStream.h"
#include "dna/CanonicalForm.h"
#include "kmers/ReadPather.h"
#include "system/System.h"
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <string>
#include <vector>

int graph_main( unsigned K, std::string const& edgesFile, std::string const& dir,
                std::string const& readsHead, std::string const& partsFile );   // ref_graph.cc (reads + parts: a.paths too)

namespace {

double now_s()
{ return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

// ---- glue restated from BuildReadQGraph48.cc:70-80 (tail finder) ----
template <unsigned K>
unsigned goodLenOf( PQVec const& pq, unsigned minQual, qvec& scratch )
{
    pq.unpack(&scratch);                       // reference decoder (feudal/PQVec.cc:129)
    unsigned run = 0;
    for ( size_t i = scratch.size(); i-- > 0; )
    {
        if ( scratch[i] < minQual ) run = 0;
        else if ( ++run == K ) return unsigned(i) + K;
    }
    return 0;
}

// ---- glue restated from BuildReadQGraph48.cc:134-190 (Kmerizer) ----
// map(): KMer<K>::kmerizeIntoEater (kmers/KMer.h:257-274) walks a read with exactly the
// initial/middle/final context rule of Kmerizer::map; the only difference is the
// len==K case (one context-free k-mer), which Kmerizer::map drops (len < K+1 -> return).
template <unsigned K>
struct RefImpl
{
    typedef KMer<K> Kmer;
    typedef KmerDictEntry<K> Entry;
    typedef KmerVec<K> KVec;

    vecbvec const* reads;
    std::vector<unsigned> const* goodLens;
    std::vector<int32_t> const* bc;     // null: no barcode test at all (K=40/60 variants)
    int64_t ignBcBelow;
    unsigned minFreq, minBC;
    KVec* out;
    std::atomic_size_t* nSolid;
    size_t mine;

    RefImpl() : reads(0), goodLens(0), bc(0), ignBcBelow(0), minFreq(0), minBC(0),
                out(0), nSolid(0), mine(0) {}
    RefImpl( RefImpl const& o ) : reads(o.reads), goodLens(o.goodLens), bc(o.bc),
        ignBcBelow(o.ignBcBelow), minFreq(o.minFreq), minBC(o.minBC), out(o.out),
        nSolid(o.nSolid), mine(0) {}
    ~RefImpl() { if ( nSolid ) *nSolid += mine; }

    template <class OItr>
    struct Eater
    {
        OItr* o; int32_t tag;
        void operator()( Kmer const& k, KMerContext kc, size_t, size_t )
        { if ( k.isRev() ) { Kmer r(k); r.rc(); **o = Entry(r,kc.rc(),tag); }
          else **o = Entry(k,kc,tag);
          ++*o; }
    };

    template <class OItr>
    void map( size_t readId, OItr oItr )
    {
        unsigned len = (*goodLens)[readId];
        if ( len < K+1 ) return;
        int32_t tag = -1;
        if ( bc && int64_t(readId) >= ignBcBelow ) tag = (*bc)[readId];
        auto beg = (*reads)[readId].begin();
        Eater<OItr> eater{&oItr,tag};
        Kmer::kmerizeIntoEater(beg,beg+len,eater,readId);
    }

    static void merge( Entry* first, Entry* last )
    {
        KMerContext all; size_t n = 0;
        for ( Entry* e = first; e != last; ++e )
        { all |= e->getKDef().getContext();
          size_t c = e->getKDef().getCount(); n += c ? c : 1; }
        first->getKDef().setContext(all);
        first->getKDef().setCount(n);            // saturates at 2^24-1 (ReadPather.h:128)
    }

    bool barcodesOK( Entry* first, Entry* last ) const
    {
        if ( !bc ) return true;
        std::vector<int32_t> seen;
        for ( Entry* e = first; e != last; ++e )
        { int32_t b = e->getTempBC();
          if ( b == -1 ) return true;
          if ( b > 0 && std::find(seen.begin(),seen.end(),b) == seen.end() )
            seen.push_back(b); }
        return seen.size() >= minBC;
    }

    void reduce( Entry* first, Entry* last )
    {
        bool ok = barcodesOK(first,last);       // before merge; merge does not touch tempBC
        merge(first,last);
        if ( ok && first->getKDef().getCount() >= minFreq )
        { ++mine; if ( out ) out->insertEntry(std::move(*first)); }
    }

    Entry* overflow( Entry* first, Entry* last )
    { if ( last-first > 1 ) merge(first,last); return first+1; }
};

// ---- glue restated from BuildReadQGraph48.cc:320-530 (EdgeBuilder, buildEdges) ----
// Unipath edges over the real KmerDict after recomputeAdjacencies: the real KMer (toSuccessor / toPredecessor / rc /
// isRev / isPalindrome), KMerContext, KDef::set and bvec::getCanonicalForm do the arithmetic; the walk is restated.
// Single-threaded (the reference's thread order only permutes the edges; buildHBVFromEdges sorts them).
template <unsigned K>
struct EdgeGlue
{
    typedef KMer<K> Kmer;
    typedef KmerDictEntry<K> Entry;
    typedef KmerDict<K> Dict;
    Dict const& dict; std::vector<bvec>& edges;
    bvec seq; std::vector<Entry const*> on;

    EdgeGlue( Dict const& d, std::vector<bvec>* e ) : dict(d), edges(*e) {}

    Entry const* find( Kmer const& k, KMerContext* ctx )
    { Entry const* r;
      if ( k.isRev() ) { r = dict.findEntryCanonical(Kmer(k).rc()); ForceAssert(r); *ctx = r->getKDef().getContext().rc(); }
      else { r = dict.findEntryCanonical(k); ForceAssert(r); *ctx = r->getKDef().getContext(); }
      return r; }

    bool canGoUp( Entry const& e )
    { KMerContext c = e.getKDef().getContext();
      if ( c.getPredecessorCount() != 1 ) return false;
      Kmer p(e); p.toPredecessor(c.getSinglePredecessor());
      if ( p.isPalindrome() ) return false;
      find(p,&c); return c.getSuccessorCount() == 1; }

    bool canGoDown( Entry const& e )
    { KMerContext c = e.getKDef().getContext();
      if ( c.getSuccessorCount() != 1 ) return false;
      Kmer n(e); n.toSuccessor(c.getSingleSuccessor());
      if ( n.isPalindrome() ) return false;
      find(n,&c); return c.getPredecessorCount() == 1; }

    void emit()
    { if ( seq.getCanonicalForm() == CanonicalForm::REV ) { seq.ReverseComplement(); std::reverse(on.begin(),on.end()); }
      EdgeID id; id.setVal(edges.size()); edges.push_back(seq);
      unsigned off = 0;
      for ( Entry const* p : on ) { ForceAssert(p->getKDef().isNull()); const_cast<KDef&>(p->getKDef()).set(id,off++); }
      seq.clear(); on.clear(); }

    void walk( Kmer const& start, KMerContext c )
    { Kmer next(start);
      while ( c.getSuccessorCount() == 1 )
      { unsigned char b = c.getSingleSuccessor(); next.toSuccessor(b);
        if ( next.isPalindrome() ) break;
        Entry const* p = find(next,&c);
        if ( c.getPredecessorCount() != 1 ) break;
        seq.push_back(b); on.push_back(p); }
      if ( seq.getCanonicalForm() == CanonicalForm::REV ) { seq.clear(); on.clear(); }    // its mirror image is built from the other end
      else emit(); }

    void fromEntry( Entry const& e )
    { bool up = false;
      if ( Kmer(e).isPalindrome() ) { seq.assign(e.begin(),e.end()); on.push_back(&e); emit(); return; }
      if ( (up = canGoUp(e)) && canGoDown(e) ) return;                                   // interior k-mer
      if ( up ) { seq.assign(e.rcbegin(),e.rcend()); on.push_back(&e); walk(Kmer(e).rc(),e.getKDef().getContext().rc()); }
      else if ( canGoDown(e) ) { seq.assign(e.begin(),e.end()); on.push_back(&e); walk(e,e.getKDef().getContext()); }
      else { seq.assign(e.begin(),e.end()); on.push_back(&e); emit(); } }

    // a k-mer still unplaced lies on a cycle without branches: the edge starts at the cycle's smallest canonical
    // k-mer, in its canonical orientation (canonicalizeCircle, :367-392)
    void circle( Entry const& first )
    { seq.assign(first.begin(),first.end()); on.push_back(&first);
      KMerContext c = first.getKDef().getContext(); Kmer k(first);
      for (;;)
      { unsigned char b = c.getSingleSuccessor(); k.toSuccessor(b);
        Entry const* p = find(k,&c);
        if ( p == &first ) break;
        ForceAssert(p->getKDef().isNull());
        seq.push_back(b); on.push_back(p); }
      size_t idx = 0;
      for ( size_t i = 1; i < on.size(); ++i ) if ( static_cast<Kmer const&>(*on[i]) < static_cast<Kmer const&>(*on[idx]) ) idx = i;
      if ( CF<K>::getForm(seq.begin(idx)) == CanonicalForm::REV )
      { seq.ReverseComplement(); std::reverse(on.begin(),on.end()); idx = seq.size()-idx-K; }
      if ( idx )
      { bvec bv; bv.assign(seq.begin(idx),seq.end()); bv.append(seq.begin(K-1),seq.begin(K+idx-1)); seq = bv;
        std::rotate(on.begin(),on.begin()+idx,on.end()); }
      emit(); }

    void run()
    { for ( auto const& hhs : dict ) for ( Entry const& e : hhs ) if ( e.getKDef().isNull() ) fromEntry(e);
      for ( auto const& hhs : dict ) for ( Entry const& e : hhs ) if ( e.getKDef().isNull() ) circle(e); }
};

// ---- glue restated from BuildReadQGraph48.cc:685-733 (Pather::path) and :593-606,622-682 (EdgeLoc / PathPart) ----
// One read against the real KmerDict (findEntry canonicalises and looks up in the reference's hopscotch set) and the
// canonical edges EdgeGlue left behind (every entry's KDef carries (edge, offset) by KDef::set): the real KMer
// (construction from an iterator, toSuccessor), CF<K>::isRC and bvec's forward / reverse-complement iterators do the
// arithmetic; what is restated is the loop: skip k-mers that are not in the dictionary (a gap part counts them), and
// from a k-mer that is, run along its edge while the bases agree (matchLen, :532-541).
// A part travels as 16 bytes: edge, offset (~offset when the read runs along the edge's reverse complement, EdgeLoc
// :596-598), k-mers covered, k-mers on the edge (0 = gap; then only `len` means anything).
struct PartOut { uint32_t edge; int32_t off; uint32_t len, elen; };

template <unsigned K>
struct PatherGlue
{
    typedef KMer<K> Kmer;
    typedef KmerDictEntry<K> Entry;
    typedef KmerDict<K> Dict;
    Dict const& dict; std::vector<bvec> const& edges;
    PatherGlue( Dict const& d, std::vector<bvec> const& e ) : dict(d), edges(e) {}

    template <class I1, class I2> static size_t agree( I1 a, I1 aEnd, I2 b, I2 bEnd )
    { size_t n = 0; while ( a != aEnd && b != bEnd && *a == *b ) { ++n; ++a; ++b; } return n; }

    void path( bvec const& read, std::vector<PartOut>* out ) const
    { out->clear();
      if ( read.size() < K ) { out->push_back(PartOut{~0u,0,unsigned(read.size()),0u}); return; }
      auto at = read.begin(); auto const stop = read.end()-K+1;
      while ( at != stop )
      { Kmer kmer(at);
        Entry const* hit = dict.findEntry(kmer);
        if ( !hit )
        { unsigned missed = 1; auto nextBase = at+K; ++at;
          while ( nextBase != read.end() )
          { kmer.toSuccessor(*nextBase); ++nextBase;
            if ( (hit = dict.findEntry(kmer)) ) break;
            ++missed; ++at; }
          out->push_back(PartOut{~0u,0,missed,0u}); }
        if ( hit )
        { KDef const& def = hit->getKDef();
          bvec const& edge = edges[def.getEdgeID().val()];
          int off = def.getEdgeOffset();
          size_t len = 1;
          bool rc = CF<K>::isRC(at,edge.begin(off));
          if ( !rc ) len += agree(at+K,read.end(),edge.begin(off)+K,edge.end());
          else
          { off = edge.size()-off;
            len += agree(at+K,read.end(),edge.rcbegin(off),edge.rcend());
            off -= K; }
          out->push_back(PartOut{unsigned(def.getEdgeID().val()),rc?~off:off,unsigned(len),unsigned(edge.size()-K+1)});
          at += len; } } }
};

struct Rec { uint64_t w0, w1; uint32_t edge, cc; int32_t bc; uint32_t pad; };

template <unsigned K>
int runDict( std::string const& head, std::string const& outdir, unsigned minQual,
             unsigned minFreq, unsigned minBC, bool useBC, unsigned nThreads, int64_t ignBcBelow, bool graph = false )
{
    typedef RefImpl<K> Impl;
    typedef typename Impl::Entry Entry;
    typedef typename Impl::Kmer Kmer;
    typedef KmerVec<K> KVec;
    typedef KmerDict<K> Dict;
    static_assert(sizeof(Entry)==32,"entry size");

    uint nt = nThreads; SetThreads(nt,False);
    vecbvec reads; reads.ReadAll((head+".fastb").c_str());
    VecPQVec quals; quals.ReadAll((head+".qualp").c_str());
    std::vector<int32_t> bc;
    if ( useBC )
    {   // DF.cc:447-452: expand bci -> per-read barcode id (0 = unbarcoded)
        vec<int64_t> bci; BinaryReader::readFile((head+".bci").c_str(),&bci);
        bc.assign(reads.size(),0);
        for ( size_t b = 0; b+1 < bci.size(); ++b )
            for ( int64_t r = bci[b]; r < bci[b+1]; ++r ) bc[r] = int32_t(b);
    }
    double t0 = now_s();
    std::vector<unsigned> goodLens(reads.size());
    { qvec scratch;
      for ( size_t r = 0; r != reads.size(); ++r )
        goodLens[r] = goodLenOf<K>(quals[r],minQual,scratch); }
    double t1 = now_s();
    size_t nKeys = 0; for ( unsigned g : goodLens ) nKeys += g;
    FILE* f = fopen((outdir+"/goodlens.u32").c_str(),"wb");
    fwrite(goodLens.data(),4,goodLens.size(),f); fclose(f);

    size_t nInst = 0;
    for ( unsigned g : goodLens ) if ( g >= K+1 ) nInst += g-K+1;

    KVec kv(0);
    double t2 = t1, t3 = t1;
    if ( nKeys )
    {
        std::atomic_size_t nSolid(0);
        { Impl impl; impl.reads=&reads; impl.goodLens=&goodLens; impl.bc = useBC?&bc:nullptr;
          impl.minFreq=minFreq; impl.minBC=minBC; impl.nSolid=&nSolid; impl.ignBcBelow=ignBcBelow;
          MapReduceEngine<Impl,Entry,typename Kmer::Hasher> mre(impl);
          if ( !mre.run(nKeys,0ul,reads.size()) ) { fprintf(stderr,"mre run1 failed\n"); return 2; } }
        t2 = now_s();
        { Impl impl; impl.reads=&reads; impl.goodLens=&goodLens; impl.bc = useBC?&bc:nullptr;
          impl.minFreq=minFreq; impl.minBC=minBC; impl.out=&kv; impl.ignBcBelow=ignBcBelow;
          typedef MapReduceEngine<Impl,Entry,typename Kmer::Hasher> MRE;
          MRE mre(impl);
          if ( !mre.run(nKeys,0ul,reads.size(),MRE::VERBOSITY::NOISY) )
          { fprintf(stderr,"mre run2 failed\n"); return 2; } }
        kv.fit();
        t3 = now_s();
        if ( kv.size() != nSolid ) { fprintf(stderr,"solid count mismatch\n"); return 3; }
    }
    // spectrum (BuildReadQGraph48.cc:192-209 semantics: histogram of counts, trailing zeros pruned)
    std::vector<int64_t> spec;
    for ( auto itr = kv.begin(); itr != kv.end(); ++itr )
    { size_t c = itr->getKDef().getCount();
      if ( spec.size() <= c ) spec.resize(c+1,0);
      spec[c]++; }
    f = fopen((outdir+"/spectrum.txt").c_str(),"w");
    for ( int64_t v : spec ) fprintf(f,"%ld\n",(long)v);
    fclose(f);
    BinaryWriter::writeFile((outdir+"/kmers.kvec").c_str(),kv);   // reference serialisation

    Dict dict(kv.size(),0.9);
    for ( auto itr = kv.begin(); itr != kv.end(); ++itr ) dict.insertEntry(*itr);
    double t4 = now_s();
    if ( minFreq > 1 ) dict.recomputeAdjacencies();               // ReadPather.h:329-364
    double t5 = now_s();

    std::vector<Rec> recs; recs.reserve(dict.size());
    for ( auto const& hhs : dict )
      for ( Entry const& e : hhs )
      { Rec r; memcpy(&r,&e,32); r.bc = -1; r.pad = 0; recs.push_back(r); }
    std::sort(recs.begin(),recs.end(),[]( Rec const& a, Rec const& b )
      { return a.w0 != b.w0 ? a.w0 < b.w0 : a.w1 < b.w1; });
    f = fopen((outdir+"/solid.bin").c_str(),"wb");
    if ( !recs.empty() ) fwrite(recs.data(),32,recs.size(),f);
    fclose(f);
    f = fopen((outdir+"/times.txt").c_str(),"w");
    fprintf(f,"reads %zu\ninstances %zu\nsolid %zu\nthreads %u\ngoodlens_s %.6f\nmr1_s %.6f\nmr2_s %.6f\ndict_s %.6f\nadj_s %.6f\n",
            reads.size(),nInst,recs.size(),nt,t1-t0,t2-t1,t3-t2,t4-t3,t5-t4);
    fclose(f);
    if ( graph )
    {   // buildEdges + buildHBVFromEdges + WriteAssemblyFiles' graph files (BuildReadQGraph48.cc:1636,1664; WriteFiles.cc:69-101)
        double t6 = now_s();
        { std::vector<bvec> found;
          { EdgeGlue<K> eg(dict,&found); eg.run(); }
          vecbvec edges; edges.reserve(found.size());            // (a MasterVec that grows by push_back corrupts its heap under g++ 11)
          for ( bvec const& b : found ) edges.push_back(b);
          edges.WriteAll((outdir+"/edges.fastb").c_str()); }
        double t7 = now_s();
        {   // pathReads, first half (Pather::path per read, BuildReadQGraph48.cc:1420-1442 calls it through HBVPather): the parts
            // of every read go to parts.bin; ref_graph.cc turns them into ReadPaths on the real digraphE
            std::vector<bvec> canon; { vecbvec e; e.ReadAll((outdir+"/edges.fastb").c_str()); for ( size_t i = 0; i != e.size(); ++i ) canon.push_back(e[i]); }
            PatherGlue<K> pg(dict,canon);
            FILE* pf = fopen((outdir+"/parts.bin").c_str(),"wb");
            uint64_t nr = reads.size(); fwrite(&nr,8,1,pf);
            std::vector<PartOut> parts;
            for ( size_t r = 0; r != reads.size(); ++r )
            { pg.path(reads[r],&parts); uint32_t n = parts.size(); fwrite(&n,4,1,pf); fwrite(parts.data(),16,n,pf); }
            fclose(pf); }
        double t8 = now_s();
        int rc = graph_main(K,outdir+"/edges.fastb",outdir+"/a."+std::to_string(K),head,outdir+"/parts.bin");
        f = fopen((outdir+"/times.txt").c_str(),"a");
        fprintf(f,"edges_s %.6f\nparts_s %.6f\nhbv_paths_s %.6f\n",t7-t6,t8-t7,now_s()-t8); fclose(f);
        return rc;
    }
    return 0;
}

template <unsigned K>
void kat( char const* s )
{
    KMer<K> k(s);
    uint64_t w[2]; memcpy(w,&k,16);
    KMer<K> r(k); r.rc();
    uint64_t v[2]; memcpy(v,&r,16);
    printf("K=%u %s w0=%016lx w1=%016lx hash=%016lx isRev=%d isPal=%d rc_w0=%016lx rc_w1=%016lx sizeof=%zu\n",
           K,s,w[0],w[1],k.hash(),int(k.isRev()),int(k.isPalindrome()),v[0],v[1],sizeof(k));
}

int mkreads( char const* in, std::string const& head )
{
    std::ifstream is(in,std::ios::binary);
    uint64_t n; is.read((char*)&n,8);
    vecbvec reads; VecPQVec quals; reads.reserve(n); quals.reserve(n);
    std::vector<unsigned char> buf;
    for ( uint64_t i = 0; i != n; ++i )
    {
        uint32_t len; is.read((char*)&len,4);
        buf.resize(2*len); is.read((char*)buf.data(),2*len);
        bvec b(len); for ( uint32_t j = 0; j != len; ++j ) b.set(j,buf[j]);
        qvec q(len); for ( uint32_t j = 0; j != len; ++j ) q[j] = buf[len+j];
        reads.push_back(b); quals.push_back(PQVec(q));      // reference encoder
    }
    reads.WriteAll((head+".fastb").c_str());
    quals.WriteAll((head+".qualp").c_str());
    return 0;
}

int rdreads( std::string const& head, char const* out )
{
    vecbvec reads; reads.ReadAll((head+".fastb").c_str());
    VecPQVec quals; quals.ReadAll((head+".qualp").c_str());
    FILE* f = fopen(out,"wb");
    uint64_t n = reads.size(); fwrite(&n,8,1,f);
    qvec q; std::vector<unsigned char> buf;
    for ( uint64_t i = 0; i != n; ++i )
    {
        uint32_t len = reads[i].size(); fwrite(&len,4,1,f);
        quals[i].unpack(&q);
        if ( q.size() != len ) { fprintf(stderr,"len mismatch read %lu\n",i); return 2; }
        buf.resize(2*len);
        for ( uint32_t j = 0; j != len; ++j ) { buf[j] = reads[i][j]; buf[len+j] = q[j]; }
        fwrite(buf.data(),1,2*len,f);
    }
    fclose(f);
    return 0;
}

// Side files DF's ingest writes next to the reads (10X/DF.cc:50-68,263-265; 10X/DfTools.cc:172-238), produced with the
// reference's containers and BinaryWriter so that our front-end's serialisation can be compared byte for byte.
struct SideDataSet { uint8_t dt; int64_t start; };                 // same layout as DataSet (10X/DfTools.h:30-45)
TRIVIALLY_SERIALIZABLE(SideDataSet);

int side( std::string const& head, std::string const& outdir, double frac0 = 1.0 )
{
    vecbvec reads0; reads0.ReadAll((head+".fastb").c_str());
    VecPQVec quals0; quals0.ReadAll((head+".qualp").c_str());
    vec<int64_t> bci0; BinaryReader::readFile((head+".bci").c_str(),&bci0);
    // LoadData (10X/DfTools.cc:99-162) for one LR input, with the reference's containers and random stream:
    // every pair draws a number, unbarcoded pairs first, then barcode by barcode, and stays if the draw <= frac
    vecbvec reads; VecPQVec quals; vec<int64_t> bci; bci.push_back(0);
    reads.reserve(reads0.size()); quals.reserve(quals0.size());
    auto decider = [frac0]() { return (1. * randomx() / RNGen::RNGEN_RAND_MAX) <= frac0; };
    for ( size_t i = 0; i < (size_t) bci0[1]; i += 2 )
        if ( decider() ) { reads.push_back(reads0[i]); reads.push_back(reads0[i+1]); quals.push_back(quals0[i]); quals.push_back(quals0[i+1]); }
    for ( size_t bc = 1; bc < bci0.size()-1; ++bc )
    {   bci.push_back( reads.size() );
        for ( size_t i = (size_t) bci0[bc]; i < (size_t) bci0[bc+1]; i += 2 )
            if ( decider() ) { reads.push_back(reads0[i]); reads.push_back(reads0[i+1]); quals.push_back(quals0[i]); quals.push_back(quals0[i+1]); } }
    bci.push_back( reads.size() );
    if ( frac0 != 1.0 )
    {   reads.WriteAll((outdir+"/frag_reads_orig.fastb").c_str()); quals.WriteAll((outdir+"/frag_reads_orig.qualp").c_str());
        BinaryWriter::writeFile((outdir+"/frag_reads_orig.bci").c_str(),bci); }
    vec<int16_t> lens(reads.size()); int maxLen = 0;
    for ( size_t i = 0; i != reads.size(); ++i ) { lens[i] = reads[i].size(); if ( lens[i] > maxLen ) maxLen = lens[i]; }
    vec<vec<vec<int64_t>>> hist( 2, vec<vec<int64_t>>( maxLen, vec<int64_t>(256,0) ) );
    qvec q; int maxQ = -1;
    for ( size_t id = 0; id != quals.size(); ++id )
    { quals[id].unpack(&q);
      for ( size_t pos = 0; pos != q.size(); ++pos ) { hist[id%2][pos][q[pos]]++; if ( q[pos] > maxQ ) maxQ = q[pos]; } }
    for ( int pos = 0; pos != maxLen; ++pos ) { hist[0][pos].resize(maxQ+1); hist[1][pos].resize(maxQ+1); }
    vec<SideDataSet> ds; SideDataSet a; memset(&a,0,sizeof a); a.dt = 2; a.start = 0; ds.push_back(a);
    a.dt = 3; a.start = bci[1]; ds.push_back(a);
    vec<String> names; names.push_back("C");
    vec<int64_t> starts(1,0);
    BinaryWriter::writeFile((outdir+"/frag_reads_orig.lens").c_str(),lens);
    BinaryWriter::writeFile((outdir+"/frag_reads_orig.qhist").c_str(),hist);
    BinaryWriter::writeFile((outdir+"/frag_reads_orig.dti").c_str(),ds);
    BinaryWriter::writeFile((outdir+"/subsam.names").c_str(),names);
    BinaryWriter::writeFile((outdir+"/subsam.starts").c_str(),starts);

    // frag_reads_orig.1000.{fastb,qualp}: the reference's own random stream (random/RNGen.cc through randomx())
    // and its own feudal writers; the glue restates 10X/DfTools.cc: LoadData draws one number per pair even at
    // LR_SELECT_FRAC = 1 (:115-117,131,147), then WriteSubSample (:32-67) keeps a pair when its draw says so or
    // when only as many pairs are left as are still wanted.
    {
        size_t pair_count = std::min<size_t>( reads.size()/2, 500 );
        double frac = static_cast<double>(pair_count)/(reads.size()/2);
        vecbvec sbases; VecPQVec squals;
        sbases.reserve(pair_count*2); squals.reserve(pair_count*2);
        for ( size_t i = 0; i+1 < reads.size() && pair_count; i += 2 )
        {
            bool take = (1. * randomx() / RNGen::RNGEN_RAND_MAX) <= frac;
            if ( take || pair_count*2 >= reads.size()-i )
            { sbases.push_back(reads[i]); sbases.push_back(reads[i+1]);
              squals.push_back(quals[i]); squals.push_back(quals[i+1]); pair_count--; }
        }
        sbases.WriteAll((outdir+"/frag_reads_orig.1000.fastb").c_str());
        squals.WriteAll((outdir+"/frag_reads_orig.1000.qualp").c_str());
    }
    return 0;
}

} // namespace

int main( int argc, char** argv )
{
    if ( argc < 2 ) { fprintf(stderr,"usage: refdrv kat|mkreads|rdreads|dict ...\n"); return 1; }
    std::string cmd = argv[1];
    if ( cmd == "kat" )
    {
        kat<48>("ACGTACGTACGTACGTACGTACGTACGTACGTTTTTTTTTTTTTTTTT");
        kat<48>("ACGTTGCAACGTTGCAACGTTGCATGCAACGTTGCAACGTTGCAACGT");
        kat<40>("ACGTACGTACGTACGTACGTACGTACGTACGTTTTTTTTT");
        kat<60>("ACGTACGTACGTACGTACGTACGTACGTACGTTTTTTTTTTTTTTTTTGGGGGGCCCCCC");
        printf("sizeof KDef=%zu Entry48=%zu Entry40=%zu Entry60=%zu KMerContext=%zu\n",
               sizeof(KDef),sizeof(KmerDictEntry<48>),sizeof(KmerDictEntry<40>),
               sizeof(KmerDictEntry<60>),sizeof(KMerContext));
        unsigned char all[256];
        for ( unsigned i = 0; i != 256; ++i )
        { KMerContext c; memcpy(&c,&(all[i]=i),1); KMerContext r = c.rc(); memcpy(&all[i],&r,1); }
        printf("ctxrc");
        for ( unsigned i = 0; i != 256; ++i ) printf(" %02x",all[i]);
        printf("\n");
        return 0;
    }
    if ( cmd == "mkreads" && argc == 4 ) return mkreads(argv[2],argv[3]);
    if ( cmd == "rdreads" && argc == 4 ) return rdreads(argv[2],argv[3]);
    if ( cmd == "side" && argc == 4 ) return side(argv[2],argv[3]);
    if ( cmd == "side" && argc == 5 ) return side(argv[2],argv[3],atof(argv[4]));
    if ( cmd == "hbv" && argc == 5 ) return graph_main(atoi(argv[2]),argv[3],argv[4],std::string(),std::string());   // hbv K edges.fastb outdir
    if ( ( cmd == "dict" || cmd == "graph" ) && ( argc == 10 || argc == 11 ) )
    {
        bool graph = cmd == "graph";                             // graph: dict + edges.fastb + a.<K>/{a.fastb,a.hbv,a.kmers,a.inv,...}
        int64_t ign = argc == 11 ? atoll(argv[10]) : 0;          // createDict's ignBcBelow (= DF's bc_start)
        unsigned K = atoi(argv[2]);
        std::string head = argv[3], outdir = argv[4];
        unsigned minQual = atoi(argv[5]), minFreq = atoi(argv[6]), minBC = atoi(argv[7]);
        bool useBC = atoi(argv[8]) != 0; unsigned nThreads = atoi(argv[9]);
        if ( K == 48 ) return runDict<48>(head,outdir,minQual,minFreq,minBC,useBC,nThreads,ign,graph);
        if ( K == 40 ) return runDict<40>(head,outdir,minQual,minFreq,minBC,useBC,nThreads,ign,graph);
        if ( K == 60 ) return runDict<60>(head,outdir,minQual,minFreq,minBC,useBC,nThreads,ign,graph);
        fprintf(stderr,"K must be 40, 48 or 60\n"); return 1;
    }
    fprintf(stderr,"bad arguments\n");
    return 1;
}
