// rm_driver.cpp -- see rm_driver.h.
#include "rm_driver.h"
#include "rm_score.h"
#include "rm_pack.h"
#include "rm_stream.h"
#include <algorithm>
#include <chrono>
#include <condition_variable>
#include <cstring>
#include <deque>
#include <mutex>
#include <thread>

namespace rma {

Replayer::Replayer( Descriptor &d, const rma_program_t &prog, FILE *out )
	: d_( d ), prog_( prog ), out_( out ), printer_( d, out )
{
	d_.score->out = out;
	for( int e = 0; e < prog.n_elems; e++ )
		if( prog.elems[ e ].re >= 0 && prog.regexes[ prog.elems[ e ].re ].loose )
			loose_ = true;
}

void Replayer::set_out( FILE *out, bool header )
{
	out_ = out;
	d_.score->out = out;
	printer_.set_out( out );
	if( !header )
		printer_.no_header();
}

void Replayer::begin()
{
	d_.score->setprog( P_BEGIN );
	d_.score->run( 0, 0, nullptr, nullptr, nullptr );
	d_.score->setprog( P_MAIN );
}

void Replayer::end()
{
	d_.score->setprog( P_END );
	d_.score->run( 0, 0, nullptr, nullptr, nullptr );
	d_.score->setprog( P_MAIN );
}

static void revcomp( std::string &s )	// mk_rcmp, rnamot.c:193-216
{
	auto cmp = []( char c ) -> char {
		switch( c ){
		case 'a' : case 'A' : return 't';
		case 'c' : case 'C' : return 'g';
		case 'g' : case 'G' : return 'c';
		case 't' : case 'T' : case 'u' : case 'U' : return 'a';
		default : return 'n';
		}
	};
	std::string	r( s.size(), 'n' );
	for( size_t i = 0, n = s.size(); i < n; i++ )
		r[ n - 1 - i ] = cmp( s[ i ] );
	s.swap( r );
}

// One candidate: restore what the search leaves in rm_descr[] at the end of the search list,
// then find_ss :373-392 -- NAME COMP POS LEN, RM_score(), print_match().
void Replayer::one_hit( const int32_t *w, const char *sid, const char *sdef, int slen, const char *sbuf, SearchStats &st )
{
	const int	ctx_off = rma_hit_ctx_off( &prog_ ), efn_off = rma_hit_efn_off( &prog_ );
	const int	comp = w[ 1 ];
	for( int e = 0; e < prog_.n_elems; e++ ){
		Strel	&s = d_.descr[ e ];
		s.matchoff = w[ RMA_HIT_HDR + 4 * e ];
		s.matchlen = w[ RMA_HIT_HDR + 4 * e + 1 ];
		s.n_mispairs = w[ RMA_HIT_HDR + 4 * e + 2 ];
		s.n_mismatches = w[ RMA_HIT_HDR + 4 * e + 3 ];
	}
	if( d_.lctx ){
		d_.lctx->matchoff = w[ ctx_off ];
		d_.lctx->matchlen = w[ ctx_off + 1 ];
	}
	if( d_.rctx ){
		d_.rctx->matchoff = w[ ctx_off + 2 ];
		d_.rctx->matchlen = w[ ctx_off + 3 ];
	}
	// Elements whose seq= the scan could only test loosely (rma_regex_t::loose: back references, letters that are not
	// acgt with iupac = 0): chk_seq() of the reference on the element's text (find_motif.c:1810-1824 -- the substring, NUL
	// terminated, through step() or, with mismatches, mm_step()).  A record that fails is no candidate of the reference's:
	// neither counted nor scored.
	if( loose_ )
		for( int e = 0; e < prog_.n_elems; e++ ){
			const int	ri = prog_.elems[ e ].re;
			if( ri < 0 || !prog_.regexes[ ri ].loose )
				continue;
			Strel	&s = d_.descr[ e ];
			chk_.assign( sbuf + s.matchoff, size_t( s.matchlen ) );
			bool	ok;
			if( s.mismatch > 0 ){
				int	n_mm = 0;
				ok = re_mm_step( *s.re, chk_.c_str(), *s.seq == '^', s.mismatch, &n_mm );
				s.n_mismatches = n_mm;
			}else
				ok = re_step( *s.re, chk_.c_str(), *s.seq == '^' );
			if( !ok )
				return;
		}
	d_.nval->pval = ( void * )sid;
	d_.cval->ival = comp;
	d_.pval->ival = comp ? slen - d_.descr[ 0 ].matchoff : d_.descr[ 0 ].matchoff + 1;
	int	len = 0;
	for( int e = 0; e < prog_.n_elems; e++ )
		len += d_.descr[ e ].matchlen;
	d_.lval->ival = len;
	st.n_candidates++;
	Ident	*h_id = nullptr;
	if( d_.score->run( comp, slen, sbuf, &h_id, prog_.n_efn_sites ? w + efn_off : nullptr ) != SA_REJECT ){
		printer_.print( sid, sdef, comp, slen, sbuf, h_id );
		st.n_hits++;
	}
}

void Replayer::replay( const std::vector<SeqRecord> &batch, const int32_t *hits, int64_t n, SearchStats &st )
{
	int	stride = rma_hit_stride( &prog_ );
	int	cur_seq = -1;
	std::string	rc;
	for( int64_t h = 0; h < n; h++ ){
		const int32_t	*w = hits + h * stride;
		int	seq = w[ 0 ], comp = w[ 1 ];
		if( seq < 0 || seq >= int( batch.size() ) )
			fail( "scanner returned a hit for sequence %d of a batch of %d.", seq, int( batch.size() ) );
		const SeqRecord	&rec = batch[ seq ];
		if( comp && seq != cur_seq ){
			rc = rec.seq;
			revcomp( rc );
			cur_seq = seq;
		}
		one_hit( w, rec.sid.c_str(), rec.sdef.c_str(), int( rec.seq.size() ), comp ? rc.c_str() : rec.seq.c_str(), st );
	}
}

void Replayer::replay_packed( const PackFile &pk, int first, const int32_t *hits, int64_t n, SearchStats &st )
{
	const int	stride = rma_hit_stride( &prog_ ), ctx_off = rma_hit_ctx_off( &prog_ );
	for( int64_t h = 0; h < n; h++ ){
		const int32_t	*w = hits + h * stride;
		const int	i = first + w[ 0 ];
		if( w[ 0 ] < 0 || i >= pk.count() )
			fail( "scanner returned a hit for sequence %d of a batch of %d.", w[ 0 ], pk.count() - first );
		const int	slen = pk.slen[ i ];
		// what the score program and print_match() read of the strand: the elements and the contexts
		int	lo = slen, hi = 0;
		auto span = [&]( int off, int len ){
			if( len > 0 ){
				lo = std::min( lo, off );
				hi = std::max( hi, off + len );
			}
		};
		for( int e = 0; e < prog_.n_elems; e++ )
			span( w[ RMA_HIT_HDR + 4 * e ], w[ RMA_HIT_HDR + 4 * e + 1 ] );
		if( d_.lctx )
			span( w[ ctx_off ], w[ ctx_off + 1 ] );
		if( d_.rctx )
			span( w[ ctx_off + 2 ], w[ ctx_off + 3 ] );
		if( text_.size() < size_t( slen ) + 1 )
			text_.resize( size_t( slen ) + 1 + size_t( slen ) / 4 );
		pk.window( i, w[ 1 ], lo, hi, text_.data() );
		one_hit( w, pk.sid( i ), pk.sdef( i ), slen, text_.data(), st );
	}
}

// ---------------------------------------------------------------- the pipelined search
// Packed batches go through two more threads: one hands them to the scanner (upload, kernels,
// copy back of the candidate records), one replays the candidates through the score program
// and the printer -- in batch order, so the output is what the serial loop prints.  While a
// batch is scanned the next one is being read and packed (rm_stream.cpp) and the one before is
// being printed.  rnamot.c:158-185 is one loop; its three parts are all that it has.
namespace {

// (laps of RNAMOTIF_TIMING: milliseconds since the first one)
double lap_clock()
{
	static const auto	t0 = std::chrono::steady_clock::now();
	return std::chrono::duration<double, std::milli>( std::chrono::steady_clock::now() - t0 ).count();
}

struct Batch {
	std::unique_ptr<PackFile>	own;	// (a slice of a packed database on disk has no copy of its own)
	const PackFile	*pk = nullptr;
	int	first = 0, count = 0;
	void	*db = nullptr;			// the batch in HBM (ScanBackend::upload_packed)
	std::vector<int32_t>	hits;
	int64_t	n_hits = 0;
};

// Replay on several threads, for score programs whose runs cannot see each other (ScoreVM::
// hit_independent: no HOLD / RELEASE, no variable carried from one candidate to the next -- SURVEY.md
// section 8e).  Every worker owns a descriptor compiled for it (the VM's variables and the element
// table the printer reads are per descriptor), takes chunks of a batch's candidates, and writes what
// the serial replay would print for them into memory; the caller's thread puts the chunks out in
// order.  A candidate that fails ends the output where the serial loop would have ended it: what
// precedes it is written, then the error is raised.  (One thread replays a gigabase's 59 000
// cloverleaf candidates in 67 ms; the kernels need 28.)
class ParallelReplayer {
public:
	ParallelReplayer( Descriptor &d, const rma_program_t &prog, FILE *out, int threads )
		: main_( d ), prog_( prog ), out_( out )
	{
		for( int t = 0; t < threads; t++ ){
			std::unique_ptr<Worker>	w( new Worker );
			w->d = compile_descriptor( d.args, &d.expanded );
			w->d->score->linkscore();
			w->d->stderr_text.clear();		// (said once, by the descriptor the program runs on)
			w->rp.reset( new Replayer( *w->d, prog, out ) );
			w->rp->set_out( out, false );
			w->rp->begin();			// BEGIN sets this VM's variables
			workers_.push_back( std::move( w ) );
		}
		for( auto &w : workers_ )
			w->th = std::thread( [ this, wp = w.get() ](){ loop( *wp ); } );
	}
	~ParallelReplayer()
	{
		{
			std::lock_guard<std::mutex>	lk( mu_ );
			quit_ = true;
		}
		cv_.notify_all();
		for( auto &w : workers_ )
			w->th.join();
	}
	void	replay_packed( Replayer &head, const PackFile &pk, int first, const int32_t *hits, int64_t n, SearchStats &st )
	{
		if( n == 0 )
			return;
		const int	stride = rma_hit_stride( &prog_ );
		const int64_t	chunk = std::max<int64_t>( 64, std::min<int64_t>( 1024, n / ( 4 * int64_t( workers_.size() ) ) + 1 ) );
		const int64_t	n_chunks = ( n + chunk - 1 ) / chunk;
		const size_t	n_pieces = size_t( n_chunks );
		std::vector<Piece>	pieces( n_pieces );
		{
			std::lock_guard<std::mutex>	lk( mu_ );
			job_ = Job{ &pk, first, hits, n, stride, chunk, n_chunks, pieces.data() };
			next_ = 0;
		}
		cv_.notify_all();
		// the pieces in order, each as soon as it is there
		for( int64_t c = 0; c < n_chunks; c++ ){
			{
				std::unique_lock<std::mutex>	lk( mu_ );
				cv_done_.wait( lk, [ & ]{ return pieces[ size_t( c ) ].ready; } );
			}
			Piece	&p = pieces[ size_t( c ) ];
			if( p.len > 0 ){
				if( head.header_pending() ){
					head.print_header( out_ );
					head.header_done();
				}
				fwrite( p.buf, 1, p.len, out_ );
			}
			free( p.buf );
			p.buf = nullptr;
			st.n_candidates += p.candidates;
			st.n_hits += p.hits;
			if( p.failed ){
				// the serial loop stops at this candidate: nothing after it is printed
				{
					std::unique_lock<std::mutex>	lk( mu_ );
					next_ = n_chunks;		// (no new chunk is begun)
					cv_done_.wait( lk, [ & ]{ return busy_ == 0; } );
					job_ = Job();
				}
				for( int64_t r = c + 1; r < n_chunks; r++ )
					free( pieces[ size_t( r ) ].buf );
				fflush( out_ );
				throw Error( p.what );
			}
		}
		std::unique_lock<std::mutex>	lk( mu_ );
		cv_done_.wait( lk, [ & ]{ return busy_ == 0; } );
		job_ = Job();
	}
private:
	struct Piece {
		char	*buf = nullptr;
		size_t	len = 0;
		int64_t	candidates = 0, hits = 0;
		bool	ready = false, failed = false;
		std::string	what;
	};
	struct Job {
		const PackFile	*pk = nullptr;
		int	first = 0;
		const int32_t	*hits = nullptr;
		int64_t	n = 0;
		int	stride = 0;
		int64_t	chunk = 0, n_chunks = 0;
		Piece	*pieces = nullptr;
	};
	struct Worker {
		std::unique_ptr<Descriptor>	d;
		std::unique_ptr<Replayer>	rp;
		std::thread	th;
	};
	void	loop( Worker &w )
	{
		for( ; ; ){
			Job	job;
			int64_t	c;
			{
				std::unique_lock<std::mutex>	lk( mu_ );
				cv_.wait( lk, [ & ]{ return quit_ || ( job_.pieces != nullptr && next_ < job_.n_chunks ); } );
				if( quit_ )
					return;
				job = job_;
				c = next_++;
				busy_++;
			}
			Piece	&p = job.pieces[ size_t( c ) ];
			const int64_t	lo = c * job.chunk, hi = std::min( job.n, lo + job.chunk );
			FILE	*fp = open_memstream( &p.buf, &p.len );
			SearchStats	st;
			bool	failed = false;
			std::string	what;
			if( fp == nullptr ){
				failed = true;
				what = "out of memory for the replay buffers";
			}else{
				w.rp->set_out( fp, false );
				try{
					w.rp->replay_packed( *job.pk, job.first, job.hits + lo * job.stride, hi - lo, st );
				}catch( Error &e ){
					failed = true;
					what = e.what();
				}
				fclose( fp );
			}
			{
				std::lock_guard<std::mutex>	lk( mu_ );
				p.candidates = st.n_candidates;
				p.hits = st.n_hits;
				p.failed = failed;
				p.what = what;
				p.ready = true;
				busy_--;
			}
			cv_done_.notify_all();
		}
	}
	Descriptor	&main_;
	const rma_program_t	&prog_;
	FILE	*out_;
	std::vector<std::unique_ptr<Worker>>	workers_;
	std::mutex	mu_;
	std::condition_variable	cv_, cv_done_;
	Job	job_;
	int64_t	next_ = 0;
	int	busy_ = 0;
	bool	quit_ = false;
};

// Three stages behind the reader, a thread each, batches handed on in order: upload (the packed words
// into HBM, on the device's upload stream), scan (kernels, ordering, copy back of the candidates),
// replay (score program and printer).  A failure anywhere sets aborting_: queued batches and the
// ones in hand are dropped, not scanned or printed -- the reference leaves at the failing candidate.
class Pipeline {
public:
	Pipeline( ScanBackend &be, Replayer &rp, const rma_program_t &prog, SearchStats &st, ParallelReplayer *par )
		: be_( be ), rp_( rp ), prog_( prog ), st_( st ), par_( par )
	{
		up_ = std::thread( [ this ](){ up_loop(); } );
		gpu_ = std::thread( [ this ](){ gpu_loop(); } );
		out_ = std::thread( [ this ](){ out_loop(); } );
	}
	~Pipeline()
	{
		{
			std::lock_guard<std::mutex>	lk( mu_ );
			closing_ = true;
			if( in_flight_ > 0 )
				aborting_ = true;	// (unwinding with batches in flight: nobody will read their output)
		}
		cv_.notify_all();
		up_.join();
		gpu_.join();
		out_.join();
		for( std::deque<Batch> *q : { &to_up_, &to_gpu_, &to_out_ } )
			for( Batch &b : *q )
				drop( b );
	}
	void	submit( Batch &&b )
	{
		std::unique_lock<std::mutex>	lk( mu_ );
		cv_.wait( lk, [ & ]{ return to_up_.size() < 2 || aborting_; } );
		rethrow( lk );
		to_up_.push_back( std::move( b ) );
		in_flight_++;
		cv_.notify_all();
	}
	void	drain()			// every batch submitted so far is printed
	{
		std::unique_lock<std::mutex>	lk( mu_ );
		cv_.wait( lk, [ & ]{ return in_flight_ == 0 || aborting_; } );
		rethrow( lk );
	}
private:
	void	rethrow( std::unique_lock<std::mutex> & )
	{
		if( failed_ ){
			failed_ = false;	// (reported once)
			throw Error( what_ );
		}
	}
	void	drop( Batch &b )
	{
		if( b.db != nullptr && be_.drop_uploaded != nullptr )
			be_.drop_uploaded( be_.self, b.db );
		b.db = nullptr;
	}
	void	fail_with( const std::string &m )
	{
		std::lock_guard<std::mutex>	lk( mu_ );
		if( !aborting_ ){
			failed_ = true;
			what_ = m;
		}
		aborting_ = true;
		cv_.notify_all();
	}
	// take the next batch of `from` (false: the pipeline is closing or aborting)
	bool	take( std::deque<Batch> &from, Batch &b )
	{
		std::unique_lock<std::mutex>	lk( mu_ );
		cv_.wait( lk, [ & ]{ return !from.empty() || closing_ || aborting_; } );
		if( aborting_ || from.empty() )
			return false;
		b = std::move( from.front() );
		from.pop_front();
		cv_.notify_all();
		return true;
	}
	// hand b on to `to` (false: aborting, b is dropped)
	bool	pass( std::deque<Batch> &to, Batch &b )
	{
		std::unique_lock<std::mutex>	lk( mu_ );
		cv_.wait( lk, [ & ]{ return to.size() < 2 || aborting_; } );
		if( aborting_ ){
			lk.unlock();
			drop( b );
			return false;
		}
		to.push_back( std::move( b ) );
		cv_.notify_all();
		return true;
	}
	void	up_loop()
	{
		for( Batch b; take( to_up_, b ); ){
			char	err[ 1024 ] = "";
			const auto	t0 = std::chrono::steady_clock::now();
			if( be_.upload_packed( be_.self, b.pk, b.first, b.count, &b.db, err, sizeof( err ) ) ){
				fail_with( std::string( "scan failed: " ) + err );
				return;
			}
			if( timing_ )
				fprintf( stderr, "[timing] upload of %d entries: %.1f ms (done at +%.1f)\n", b.count,
					std::chrono::duration<double, std::milli>( std::chrono::steady_clock::now() - t0 ).count(), lap_clock() );
			if( !pass( to_gpu_, b ) )
				return;
		}
	}
	void	gpu_loop()
	{
		for( Batch b; take( to_gpu_, b ); ){
			char	err[ 1024 ] = "";
			const int32_t	*hits = nullptr;
			const auto	t0 = std::chrono::steady_clock::now();
			void	*db = b.db;
			b.db = nullptr;		// (scan_uploaded takes it, whatever it returns)
			if( be_.scan_uploaded( be_.self, db, &hits, &b.n_hits, err, sizeof( err ) ) ){
				fail_with( std::string( "scan failed: " ) + err );
				return;
			}
			b.hits.assign( hits, hits + b.n_hits * rma_hit_stride( &prog_ ) );	// (the scanner's buffer is its next scan's)
			if( timing_ )
				fprintf( stderr, "[timing] scan of %d entries: %.1f ms, %lld candidates (done at +%.1f)\n", b.count,
					std::chrono::duration<double, std::milli>( std::chrono::steady_clock::now() - t0 ).count(), ( long long )b.n_hits, lap_clock() );
			if( !pass( to_out_, b ) )
				return;
		}
	}
	void	out_loop()
	{
		for( Batch b; take( to_out_, b ); ){
			const auto	t0 = std::chrono::steady_clock::now();
			try{
				if( par_ != nullptr )
					par_->replay_packed( rp_, *b.pk, b.first, b.hits.data(), b.n_hits, st_ );
				else
					rp_.replay_packed( *b.pk, b.first, b.hits.data(), b.n_hits, st_ );
			}catch( Error &e ){
				fail_with( e.what() );
				return;
			}
			if( timing_ )
				fprintf( stderr, "[timing] replay of %lld candidates: %.1f ms (done at +%.1f)\n", ( long long )b.n_hits,
					std::chrono::duration<double, std::milli>( std::chrono::steady_clock::now() - t0 ).count(), lap_clock() );
			if( b.own ){		// (a batch parsed from text: its words serve the next one)
				recycle_words( std::move( b.own->codes ) );
				recycle_words( std::move( b.own->amask ) );
			}
			b = Batch();		// (the batch's own pack goes before the count says it is done)
			std::lock_guard<std::mutex>	lk( mu_ );
			if( in_flight_ > 0 )
				in_flight_--;
			cv_.notify_all();
		}
	}
	ScanBackend	&be_;
	Replayer	&rp_;
	const rma_program_t	&prog_;
	SearchStats	&st_;
	ParallelReplayer	*par_;
	std::mutex	mu_;
	std::condition_variable	cv_;
	std::deque<Batch>	to_up_, to_gpu_, to_out_;
	int	in_flight_ = 0;
	bool	closing_ = false, failed_ = false, aborting_ = false;
	const bool	timing_ = getenv( "RNAMOTIF_TIMING" ) != nullptr;
	std::string	what_;
	std::thread	up_, gpu_, out_;
};

int parser_threads()
{
	if( const char *t = getenv( "RNAMOTIF_THREADS" ) )
		return std::max( 1, atoi( t ) );
	const unsigned	hw = std::thread::hardware_concurrency();
	return int( std::max( 1u, std::min( 32u, hw > 3 ? hw - 2 : 1u ) ) );
}

}	// namespace

int run_search( Descriptor &d, const rma_program_t &prog, ScanBackend &be, FILE *out,
	int64_t batch_bases, SearchStats *stats )
{
	SearchStats	st;
	( void )lap_clock();		// (origin of the laps)
	Replayer	rp( d, prog, out );
	rp.begin();
	bool	use_stdin = d.args.dbfnames.empty();
	size_t	nfiles = use_stdin ? 1 : d.args.dbfnames.size();
	std::vector<SeqRecord>	batch;
	int64_t	in_batch = 0;
	char	err[ 1024 ];
	int	ecnt = 0;
	const int	show_progress = d.int_global( "show_progress", 0 );
	// (RNAMOTIF_SERIAL: the one-thread loop over text, as the reference has it)
	const bool	piped = be.upload_packed != nullptr && be.scan_uploaded != nullptr && getenv( "RNAMOTIF_SERIAL" ) == nullptr;
	// (declared before the pipeline: its threads upload from and print from these packs until they are
	// joined, so the packs must be the last to go when an error unwinds this function)
	std::vector<std::unique_ptr<PackFile>>	packs;		// (alive until the last batch is printed)
	std::unique_ptr<ParallelReplayer>	par;
	std::unique_ptr<Pipeline>	pl;
	if( piped ){
		// candidates that cannot see each other's runs of the score program are replayed on several threads
		const char	*rt = getenv( "RNAMOTIF_REPLAY_THREADS" );
		const unsigned	hw = std::thread::hardware_concurrency();
		const int	threads = rt ? atoi( rt ) : int( std::max( 1u, std::min( 8u, hw / 2 ) ) );
		std::string	why;
		if( threads > 1 && d.score->hit_independent( &why ) ){
			// (a worker that cannot be set up is no reason to give up a search one thread can replay)
			try{
				par.reset( new ParallelReplayer( d, prog, out, threads ) );
			}catch( const Error &e ){
				par.reset();
				why = std::string( "workers could not be set up: " ) + e.what();
			}
		}
		if( getenv( "RNAMOTIF_TIMING" ) )
			fprintf( stderr, "[timing] replay on %d thread(s)%s%s\n", par ? threads : 1, par || threads <= 1 ? "" : ": ", par || threads <= 1 ? "" : why.c_str() );
		pl.reset( new Pipeline( be, rp, prog, st, par.get() ) );
	}
	auto flush = [&](){
		if( batch.empty() )
			return;
		if( pl )
			pl->drain();		// text batches are printed by this thread: after what is in flight
		std::vector<const char *>	seqs;
		std::vector<int32_t>	slens;
		for( const SeqRecord &r : batch ){
			seqs.push_back( r.seq.c_str() );
			slens.push_back( int32_t( r.seq.size() ) );
		}
		const int32_t	*hits = nullptr;
		int64_t	n_hits = 0;
		err[ 0 ] = '\0';
		if( be.scan( be.self, seqs.data(), slens.data(), int( batch.size() ), &hits, &n_hits, err, sizeof( err ) ) )
			fail( "scan failed: %s", err );
		rp.replay( batch, hits, n_hits, st );
		batch.clear();
		in_batch = 0;
	};
	// rnamot.c:160-176: the entry counter, and show_progress's line every so many entries
	auto tick = [&]( const char *sid ){
		ecnt++;
		if( show_progress > 0 && ecnt % show_progress == 0 )
			fprintf( stderr, "%s: %7d: %s\n", d.args.argv0.c_str(), ecnt, sid );
	};
	// the EOF of file f: DB_fnext (dbutil.c:12-40) opens the next file before the main loop counts
	// the EOF as an entry (rnamot.c:160-168); a file that cannot be read ends the run uncounted
	bool	stop = false;
	auto next_file = [&]( size_t f ){
		if( use_stdin || f + 1 >= nfiles )
			return;
		FILE	*t = fopen( d.args.dbfnames[ f + 1 ].c_str(), "r" );
		if( t == nullptr ){
			fprintf( stderr, "DB_fnext: can't read seq file '%s'.\n", d.args.dbfnames[ f + 1 ].c_str() );
			stop = true;
			return;
		}
		fclose( t );
		tick( "" );
	};
	auto submit = [&]( std::unique_ptr<PackFile> own, const PackFile *pk, int first, int count ){
		int64_t	bases = 0;
		for( int i = 0; i < count; i++ ){
			tick( pk->sid( first + i ) );
			bases += pk->slen[ first + i ];
		}
		st.n_seqs += count;
		st.n_bases += bases;
		Batch	b;
		b.own = std::move( own );
		b.pk = pk;
		b.first = first;
		b.count = count;
		pl->submit( std::move( b ) );
	};
	// A packed database (rm_pack.h) takes the place of a text file: same entries, same order.
	auto scan_pack = [&]( const std::string &path ){
		packs.emplace_back( new PackFile );
		PackFile	&pk = *packs.back();
		std::string	perr;
		const auto	t0 = std::chrono::steady_clock::now();
		// (tables and names now, the packed bases batch by batch while the batches before are searched)
		if( !pk.open( path, perr ) )
			fail( "%s", perr.c_str() );
		if( getenv( "RNAMOTIF_TIMING" ) )
			fprintf( stderr, "[timing] pack opened: %.1f ms\n", std::chrono::duration<double, std::milli>( std::chrono::steady_clock::now() - t0 ).count() );
		int	first = 0;
		while( first < pk.count() ){
			int	count = 0;
			int64_t	bases = 0;
			while( first + count < pk.count() && ( count == 0 || bases < batch_bases ) ){
				bases += pk.slen[ first + count ];
				count++;
			}
			const auto	t1 = std::chrono::steady_clock::now();
			if( !pk.ensure( first + count, perr ) )
				fail( "%s", perr.c_str() );
			if( getenv( "RNAMOTIF_TIMING" ) )
				fprintf( stderr, "[timing] batch of %d entries read from the pack: %.1f ms (done at +%.1f)\n", count,
					std::chrono::duration<double, std::milli>( std::chrono::steady_clock::now() - t1 ).count(), lap_clock() );
			if( pl ){
				submit( nullptr, &pk, first, count );
				first += count;
				continue;
			}
			// (a backend without a packed entry point: through text)
			for( int i = 0; i < count; i++ )
				tick( pk.sid( first + i ) );
			st.n_seqs += count;
			st.n_bases += bases;
			std::vector<SeqRecord>	recs;
			recs.resize( size_t( count ) );
			std::vector<const char *>	seqs;
			std::vector<int32_t>	slens;
			for( int i = 0; i < count; i++ ){
				recs[ i ].sid = pk.sid( first + i );
				recs[ i ].sdef = pk.sdef( first + i );
				recs[ i ].seq = pk.unpack( first + i );
				seqs.push_back( recs[ i ].seq.c_str() );
				slens.push_back( pk.slen[ first + i ] );
			}
			const int32_t	*hits = nullptr;
			int64_t	n_hits = 0;
			err[ 0 ] = '\0';
			if( be.scan( be.self, seqs.data(), slens.data(), count, &hits, &n_hits, err, sizeof( err ) ) )
				fail( "scan failed: %s", err );
			rp.replay( recs, hits, n_hits, st );
			first += count;
		}
	};
	for( size_t f = 0; f < nfiles && !stop; f++ ){
		FILE	*fp = stdin;
		if( !use_stdin && PackFile::is_pack( d.args.dbfnames[ f ] ) ){
			flush();
			scan_pack( d.args.dbfnames[ f ] );
			next_file( f );
			continue;
		}
		long	resume_at = 0;
		if( !use_stdin && pl && seq_format_of( d.args.dbfmt ) == FMT_FASTN ){
			// a FASTA file: read, packed and scanned in parallel as far as it is regular
			FastaStream	fs;
			fs.keep_mapping();		// (see there: unmapping stalls the batches still in flight)
			if( fs.open( d.args.dbfnames[ f ], d.args.maxslen, parser_threads() ) ){
				flush();
				auto	t0 = std::chrono::steady_clock::now();
				while( std::unique_ptr<PackFile> pk = fs.next( batch_bases ) ){
					if( getenv( "RNAMOTIF_TIMING" ) ){
						fprintf( stderr, "[timing] batch of %d entries read and packed: %.1f ms\n", pk->count(),
							std::chrono::duration<double, std::milli>( std::chrono::steady_clock::now() - t0 ).count() );
						t0 = std::chrono::steady_clock::now();
					}
					const PackFile	*p = pk.get();
					submit( std::move( pk ), p, 0, p->count() );
				}
				if( fs.stopped_at() < 0 ){
					next_file( f );
					continue;
				}
				resume_at = long( fs.stopped_at() );	// an entry the reader has something to say about
			}
		}
		if( !use_stdin ){
			fp = fopen( d.args.dbfnames[ f ].c_str(), "r" );
			if( fp == nullptr ){
				// DB_fnext, dbutil.c:12-40: report and stop (the first file; later ones are tried at the EOF before them)
				fprintf( stderr, "DB_fnext: can't read seq file '%s'.\n", d.args.dbfnames[ f ].c_str() );
				break;
			}
			if( resume_at > 0 )
				fseek( fp, resume_at, SEEK_SET );
		}
		FastaReader	rd( fp, d.args.maxslen, seq_format_of( d.args.dbfmt ) );
		SeqRecord	rec;
		for( ; ; ){
			if( !rd.next( rec ) ){
				next_file( f );
				break;
			}
			tick( rec.sid.c_str() );
			st.n_seqs++;
			st.n_bases += int64_t( rec.seq.size() );
			in_batch += int64_t( rec.seq.size() );
			batch.push_back( std::move( rec ) );
			if( in_batch >= batch_bases )
				flush();
		}
		if( fp != stdin )
			fclose( fp );
	}
	flush();
	if( pl ){
		pl->drain();
		if( getenv( "RNAMOTIF_TIMING" ) )
			fprintf( stderr, "[timing] pipeline drained at +%.1f\n", lap_clock() );
		pl.reset();
		par.reset();
		if( getenv( "RNAMOTIF_TIMING" ) )
			fprintf( stderr, "[timing] pipeline threads joined at +%.1f\n", lap_clock() );
	}
	rp.end();
	if( stats )
		*stats = st;
	// The packed databases are left to the end of the process (the command line program leaves with
	// _exit() after its last line, rm_main.cpp): returning a gigabase of touched pages to the system
	// page by page took 55 ms of a 160 ms search, for memory the exit hands back wholesale.
	for( std::unique_ptr<PackFile> &pk : packs )
		( void )pk.release();
	return 0;
}

}	// namespace rma
