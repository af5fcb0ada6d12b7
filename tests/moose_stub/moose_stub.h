// TEST INFRASTRUCTURE, not product code and not a build of the reference: the smallest stand-ins for the MOOSE / libMesh / Marlin
// base classes that the files of marlin_plugin/ derive from and call, so that those files -- the reference-side binding of
// include/marlin_hip.h -- are COMPILED AS THEY ARE and EXECUTED on the GPU box (tests/test_moose_shim_gpu.py), where no MOOSE exists.
// Nothing here evaluates a transform or a solver step; tensors are real libTorch tensors on the HIP device, as in Marlin.
//
// Behaviour written from (paths relative to idaholab/marlin):
//   include/tensor_solver/TensorSolver.h:17-62, src/tensor_solver/TensorSolver.C:41-109   members, getBufferOld, the substep loop
//   include/tensor_solver/SplitOperatorBase.h:17-37, src/tensor_solver/SplitOperatorBase.C:13-64   Variable, getVariables
//   include/tensor_computes/TensorOperatorBase.h:26-151, TensorOperator.h:17-45            buffer getters, _time = subTime()
//   include/tensor_buffers/TensorBuffer.h:17-118                                            advanceState of one buffer
//   src/problems/TensorProblem.C:160-190, 451-472                                           execute(TIMESTEP_BEGIN), advanceState, timeStep() <= 1
//   src/tensor_computes/ComputeGroup.C:61-84                                                computes of a group run in order
//   MOOSE: InputParameters (addParam / addRequiredParam / addRangeCheckedParam / set / get), MooseObject (getParam, isParamValid,
//   paramError, name, comm, _console), registerMooseObject, mooseError, TransientBase::incrementStepOrReject / takeStep
//   (t_step += 1, advanceState, dt_old = dt).
// The same-named forwarding headers next to this file (SplitOperatorBase.h, TensorOperator.h, ...) only include this one.
#pragma once

#include <torch/torch.h>

#include <fcntl.h>
#include <sys/mman.h>
#include <unistd.h>

#include <algorithm>
#include <cmath>
#include <any>
#include <array>
#include <atomic>
#include <cstring>
#include <cstdint>
#include <functional>
#include <iostream>
#include <map>
#include <memory>
#include <set>
#include <sstream>
#include <stdexcept>
#include <string>
#include <typeinfo>
#include <vector>

using Real = double;
typedef std::string TensorComputeName;
typedef std::string TensorInputBufferName;
typedef std::string TensorOutputBufferName;

namespace moose_stub
{
template <typename... A>
std::string
cat(A &&... a)
{
  std::ostringstream s;
  (s << ... << a);
  return s.str();
}
/// where the tensors of a run live (the reference: MooseTensor::floatTensorOptions(), src/utils/MarlinUtils.C:100-123)
inline torch::Device &
device()
{
  static torch::Device d(torch::kCPU);
  return d;
}

template <typename T>
struct FromString;
template <>
struct FromString<std::string>
{
  static std::string get(const std::string & s) { return s; }
};
template <>
struct FromString<bool>
{
  static bool get(const std::string & s) { return s == "true" || s == "1" || s == "TRUE"; }
};
template <>
struct FromString<double>
{
  static double get(const std::string & s) { return std::stod(s); }
};
template <>
struct FromString<unsigned int>
{
  static unsigned int get(const std::string & s) { return (unsigned int)std::stoul(s); }
};
template <>
struct FromString<std::size_t>
{
  static std::size_t get(const std::string & s) { return std::stoull(s); }
};
template <typename T>
struct FromString<std::vector<T>>
{
  static std::vector<T> get(const std::string & s)
  {
    std::vector<T> v;
    std::istringstream in(s);
    std::string tok;
    while (in >> tok)
      v.push_back(FromString<T>::get(tok));
    return v;
  }
};
/// how a declared parameter takes a value from text (MooseEnum keeps its list of names: see below)
template <typename T>
void
assign(T & v, const std::string & text)
{
  v = FromString<T>::get(text);
}
} // namespace moose_stub

struct MooseStubError : std::runtime_error
{
  using std::runtime_error::runtime_error;
};

template <typename... A>
[[noreturn]] void
mooseError(A &&... a)
{
  throw MooseStubError(moose_stub::cat(std::forward<A>(a)...));
}

namespace moose_stub
{
/// the "MPI job" of a multi-rank run of the driver: rank processes forked by shim-driver share one POSIX shared-memory page
struct World
{
  unsigned int size = 1, rank = 0;
  bool pencil = false;   // parallel_mode = FFT_PENCIL (FFT_SLAB otherwise)
  struct Page
  {
    std::atomic<unsigned int> seq, arrived;
    unsigned int len;
    char text[3072];
  } * page = nullptr;
};
inline World &
world()
{
  static World w;
  return w;
}
/// join the job `name` as rank `rank` of `size` (creates / maps the page; rank 0 clears it before the others can see it is ready)
inline void
joinWorld(const std::string & name, unsigned int size, unsigned int rank)
{
  auto & w = world();
  w.size = size;
  w.rank = rank;
  if (size == 1)
    return;
  const int fd = shm_open(("/" + name).c_str(), O_CREAT | O_RDWR, 0600);
  if (fd < 0 || ftruncate(fd, sizeof(World::Page)) != 0)
    throw std::runtime_error("moose_stub: cannot create the job page");
  void * m = mmap(nullptr, sizeof(World::Page), PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
  close(fd);
  if (m == MAP_FAILED)
    throw std::runtime_error("moose_stub: cannot map the job page");
  w.page = static_cast<World::Page *>(m); // (a fresh segment is zero-filled: seq = arrived = 0)
}
}

namespace libMesh
{
namespace Parallel
{
/// size / rank of the job and the one collective the shim uses (the job-name broadcast of HipDomain); one rank unless the driver was
/// started with nranks = P
class Communicator
{
public:
  unsigned int size() const { return moose_stub::world().size; }
  unsigned int rank() const { return moose_stub::world().rank; }
  /// broadcast from rank 0
  void broadcast(std::string & s) const
  {
    auto & w = moose_stub::world();
    if (w.size == 1)
      return;
    auto * p = w.page;
    const unsigned int round = p->seq.load() / 2; // every rank has finished the previous broadcast before anyone starts the next
    int spins = 0;
    auto wait_for = [&](unsigned int v) {
      while (p->seq.load() != v)
      {
        usleep(50);
        if (++spins > 2400000) // two minutes: a rank has died
          throw std::runtime_error("moose_stub: broadcast timed out (a rank is gone)");
      }
    };
    if (w.rank == 0)
    {
      p->len = (unsigned int)std::min<std::size_t>(s.size(), sizeof(p->text));
      std::memcpy(p->text, s.data(), p->len);
      p->seq.store(2 * round + 1);
    }
    else
    {
      wait_for(2 * round + 1);
      s.assign(p->text, p->len);
    }
    if (p->arrived.fetch_add(1) + 1 == w.size)
    {
      p->arrived.store(0);
      p->seq.store(2 * round + 2);
    }
    else
      wait_for(2 * round + 2);
  }
};
}
}

class RealVectorValue
{
public:
  Real operator()(unsigned int i) const { return _v[i]; }
  Real & operator()(unsigned int i) { return _v[i]; }

private:
  std::array<Real, 3> _v{{0, 0, 0}};
};

namespace MooseTensor
{
inline const torch::TensorOptions
floatTensorOptions()
{
  return torch::TensorOptions().dtype(torch::kFloat64).device(moose_stub::device());
}
inline const torch::TensorOptions
complexFloatTensorOptions()
{
  return torch::TensorOptions().dtype(torch::kComplexDouble).device(moose_stub::device());
}
}

/// MooseEnum("X=0 Y=1 Z=2"[, "default"]): a named choice that converts to its integer
class MooseEnum
{
public:
  MooseEnum() = default;
  MooseEnum(const std::string & options, const std::string & selected = "")
  {
    std::istringstream in(options);
    std::string tok;
    int next = 0;
    while (in >> tok)
    {
      const auto eq = tok.find('=');
      const int id = eq == std::string::npos ? next : std::stoi(tok.substr(eq + 1));
      _names.emplace_back(tok.substr(0, eq), id);
      next = id + 1;
    }
    if (!selected.empty())
      select(selected);
  }
  void select(const std::string & name)
  {
    for (const auto & n : _names)
      if (n.first == name)
      {
        _current = n.second;
        _valid = true;
        return;
      }
    mooseError("'", name, "' is not one of the values of this MooseEnum");
  }
  bool isValid() const { return _valid; }
  operator int() const { return _current; }

private:
  std::vector<std::pair<std::string, int>> _names;
  int _current = -1;
  bool _valid = false;
};
namespace moose_stub
{
template <>
inline void
assign<MooseEnum>(MooseEnum & v, const std::string & text)
{
  v.select(text);
}
}

class InputParameters
{
public:
  void addClassDescription(const std::string & d) { _description = d; }
  void registerBase(const std::string &) {}
  template <typename T>
  void addParam(const std::string & name, const T & value, const std::string & doc)
  {
    declare<T>(name, doc, false);
    _values[name] = value;
  }
  template <typename T>
  void addParam(const std::string & name, const std::string & doc)
  {
    declare<T>(name, doc, false);
  }
  template <typename T>
  void addRequiredParam(const std::string & name, const std::string & doc)
  {
    declare<T>(name, doc, true);
  }
  /// required, but declared with a prototype value that carries the choices (MooseEnum)
  template <typename T>
  void addRequiredParam(const std::string & name, const T & prototype, const std::string & doc)
  {
    declare<T>(name, doc, true);
    _values[name] = prototype;
  }
  template <typename T>
  void addRangeCheckedParam(const std::string & name, const T & value, const std::string & range, const std::string & doc)
  {
    declare<T>(name, doc + " [" + range + "]", false);
    _values[name] = value;
  }
  template <typename T>
  T & set(const std::string & name)
  {
    auto & slot = _values[name];
    if (!slot.has_value() || slot.type() != typeid(T))
      slot = T{};
    return *std::any_cast<T>(&slot);
  }
  template <typename T>
  const T & get(const std::string & name) const
  {
    auto it = _values.find(name);
    if (it == _values.end() || !it->second.has_value())
      mooseError("parameter '", name, "' has no value");
    const T * p = std::any_cast<T>(&it->second);
    if (!p)
      mooseError("parameter '", name, "' was declared with another type than the one it is read with");
    return *p;
  }
  bool isParamValid(const std::string & name) const
  {
    auto it = _values.find(name);
    return it != _values.end() && it->second.has_value();
  }
  bool declared(const std::string & name) const { return _parse.count(name) != 0; }
  /// the input-file side: a value arrives as text and is converted to the declared type (unknown names are an error, as in MOOSE)
  void setFromString(const std::string & name, const std::string & text)
  {
    auto it = _parse.find(name);
    if (it == _parse.end())
      mooseError("unused parameter '", name, "'");
    it->second(_values[name], text);
    _set_by_user.insert(name);
  }
  void checkRequired(const std::string & object) const
  {
    for (const auto & r : _required)
      if (!_set_by_user.count(r))
        mooseError(object, ": missing required parameter '", r, "'");
  }

private:
  template <typename T>
  void declare(const std::string & name, const std::string & doc, bool required)
  {
    _doc[name] = doc;
    _parse[name] = [](std::any & slot, const std::string & text)
    {
      T v = (slot.has_value() && slot.type() == typeid(T)) ? *std::any_cast<T>(&slot) : T{};
      moose_stub::assign(v, text);
      slot = v;
    };
    if (required)
      _required.insert(name);
  }
  std::string _description;
  std::map<std::string, std::any> _values;
  std::map<std::string, std::string> _doc;
  std::map<std::string, std::function<void(std::any &, const std::string &)>> _parse;
  std::set<std::string> _required, _set_by_user;
};

class MooseObject
{
public:
  MooseObject(const InputParameters & parameters) : _pars(parameters), _console(std::cout) {}
  virtual ~MooseObject() = default;
  const std::string & name() const { return _pars.get<std::string>("_object_name"); }
  template <typename T>
  const T & getParam(const std::string & n) const
  {
    return _pars.get<T>(n);
  }
  bool isParamValid(const std::string & n) const { return _pars.isParamValid(n); }
  template <typename... A>
  [[noreturn]] void paramError(const std::string & param, A &&... a) const
  {
    mooseError(name(), ": parameter '", param, "': ", std::forward<A>(a)...);
  }
  template <typename... A>
  void paramWarning(const std::string & param, A &&... a) const
  {
    std::cerr << "*** Warning ***\n" << moose_stub::cat(name(), ": parameter '", param, "': ", std::forward<A>(a)...) << "\n";
  }
  const libMesh::Parallel::Communicator & comm() const
  {
    static libMesh::Parallel::Communicator c;
    return c;
  }

protected:
  const InputParameters _pars;
  std::ostream & _console;
};

/// registerMooseObject: the factory of the test driver (type name -> validParams + constructor)
struct MooseStubFactory
{
  struct Entry
  {
    std::function<InputParameters()> valid_params;
    std::function<std::shared_ptr<MooseObject>(const InputParameters &)> build;
  };
  static std::map<std::string, Entry> & registry()
  {
    static std::map<std::string, Entry> r;
    return r;
  }
  template <typename T>
  static int add(const std::string & type)
  {
    registry()[type] = Entry{[]() { return T::validParams(); },
                             [](const InputParameters & p) -> std::shared_ptr<MooseObject> { return std::make_shared<T>(p); }};
    return 0;
  }
};
#define registerMooseObject(app, classname)                                                                            \
  static const int moose_stub_registered_##classname = MooseStubFactory::add<classname>(#classname)

/// include/actions/DomainAction.h:31-69 -- the getters the shim reads (+ partitionPencils when the job runs FFT_PENCIL, see below).  parallel_mode = NONE (DomainAction.C:268-296: r2c on the last
/// axis) with one rank; with several ranks parallel_mode = FFT_SLAB as DomainAction::partitionSlabs does it (:510-566): the real space
/// is split along y, the reciprocal space along x, every axis transforms c2c (:278-280), equal weights through partitionHepler
/// (DomainAction.h:247-280); getLocalBounds hands out _local_begin / _local_end, i.e. the RECIPROCAL x range on axis 0 and the real y
/// range on axis 1 (:524-533, 1544-1556).
class DomainAction
{
public:
  DomainAction(unsigned int dim, std::array<int64_t, 3> n, std::array<Real, 3> lo, std::array<Real, 3> hi) : _dim(dim), _n(n)
  {
    const auto & w = moose_stub::world();
    _n_rank = w.size;
    _rank = w.rank;
    for (unsigned int d = 0; d < 3; ++d)
    {
      if (d >= dim)
        _n[d] = 1;
      _min(d) = lo[d];
      _max(d) = hi[d];
    }
    _n_reciprocal = _n;
    if (_n_rank == 1)
      _n_reciprocal[dim - 1] = _n[dim - 1] / 2 + 1;
    _n_local = _n;
    _n_reciprocal_local = _n_reciprocal;
    for (unsigned int d = 0; d < 3; ++d)
    {
      _begin[d].assign(_n_rank, 0);
      _end[d].assign(_n_rank, _n[d]);
    }
    auto split = [](int64_t total, unsigned int parts)   // partitionHepler with unit weights (DomainAction.h:247-280)
    {
      std::vector<int64_t> counts(parts);
      int64_t remaining = parts;
      for (unsigned int r = 0; r < parts; ++r)
      {
        int64_t c = std::max<int64_t>(total / remaining, 1);
        if (r + 1 == parts)
          c = total;
        counts[r] = c;
        total -= c;
        remaining -= 1;
      }
      return counts;
    };
    auto offset = [](const std::vector<int64_t> & counts, unsigned int i)
    {
      int64_t o = 0;
      for (unsigned int k = 0; k < i; ++k)
        o += counts[k];
      return o;
    };
    if (_n_rank > 1 && w.pencil)
    {
      // DomainAction::partitionPencils (DomainAction.C:568-698): r2c on x (:282-284), the real space split along y (rank % Py) and z
      // (rank / Py), the reciprocal space along kx (rank % Py) and ky (rank / Py); Py x Pz = the most balanced factorisation of the ranks
      if (dim < 3)
        mooseError("Dimension must be 3 for pencil decomposition.");
      _n_reciprocal = _n;
      _n_reciprocal[0] = _n[0] / 2 + 1;
      unsigned int Py = 0, Pz = 0, best = ~0u;
      auto consider = [&](unsigned int px, unsigned int pz)
      {
        if (px < 2 || pz < 2 || px > _n[1] || px > _n_reciprocal[0] || pz > _n[2] || pz > _n[1])
          return;
        const unsigned int cost = px > pz ? px - pz : pz - px;
        if (Py == 0 || cost < best)
        {
          Py = px;
          Pz = pz;
          best = cost;
        }
      };
      const unsigned int max_divisor = std::max(2u, (unsigned int)std::sqrt((double)_n_rank));
      for (unsigned int d = 2; d <= max_divisor; ++d)
        if (_n_rank % d == 0)
        {
          consider(d, _n_rank / d);
          consider(_n_rank / d, d);
        }
      if (Py == 0)
        mooseError("FFT_PENCIL requires factoring the number of MPI ranks into two integers greater than one that fit the domain (ranks = ",
                   _n_rank, "). Use FFT_SLAB or adjust the rank count.");
      const auto yc = split(_n[1], Py), zc = split(_n[2], Pz), kxc = split(_n_reciprocal[0], Py), kyc = split(_n_reciprocal[1], Pz);
      for (unsigned int r = 0; r < _n_rank; ++r)
      {
        const unsigned int py = r % Py, pz = r / Py;
        _begin[1][r] = offset(yc, py);
        _end[1][r] = _begin[1][r] + yc[py];
        _begin[2][r] = offset(zc, pz);
        _end[2][r] = _begin[2][r] + zc[pz];
      }
      _n_local[1] = yc[_rank % Py];
      _n_local[2] = zc[_rank / Py];
      _n_reciprocal_local = {{kxc[_rank % Py], kyc[_rank / Py], _n_reciprocal[2]}};
    }
    else if (_n_rank > 1)
    {
      if (dim < 2)
        mooseError("Dimension must be 2 or 3 for slab decomposition.");
      const int64_t totals[2] = {_n_reciprocal[0], _n[1]};
      for (unsigned int d = 0; d < 2; ++d)
      {
        const auto counts = split(totals[d], _n_rank);
        for (unsigned int r = 0; r < _n_rank; ++r)
        {
          _begin[d][r] = offset(counts, r);
          _end[d][r] = _begin[d][r] + counts[r];
        }
      }
      _n_local[1] = _end[1][_rank] - _begin[1][_rank];
      _n_reciprocal_local[0] = _end[0][_rank] - _begin[0][_rank];
    }
    _shape_store.assign(_n_local.begin(), _n_local.begin() + dim);
    _reciprocal_store.assign(_n_reciprocal_local.begin(), _n_reciprocal_local.begin() + dim);
    _shape = _shape_store;
    _reciprocal_shape = _reciprocal_store;
  }
  DomainAction(const DomainAction &) = delete; // (_shape points into this object)
  const unsigned int & getDim() const { return _dim; }
  bool isRealSpaceMode() const { return false; }
  const std::array<int64_t, 3> & getGridSize() const { return _n; }
  const std::array<int64_t, 3> & getReciprocalGridSize() const { return _n_reciprocal; }
  const std::array<int64_t, 3> & getLocalGridSize() const { return _n_local; }
  const std::array<int64_t, 3> & getLocalReciprocalGridSize() const { return _n_reciprocal_local; }
  const RealVectorValue & getDomainMin() const { return _min; }
  const RealVectorValue & getDomainMax() const { return _max; }
  const torch::IntArrayRef & getShape() const { return _shape; }
  const torch::IntArrayRef & getReciprocalShape() const { return _reciprocal_shape; }
  bool isParallelFFT() const { return _n_rank > 1; }
  void getLocalBounds(unsigned int rank, std::array<int64_t, 3> & begin, std::array<int64_t, 3> & end) const
  {
    if (rank >= _n_rank)
      mooseError("Requested local bounds for invalid rank ", rank, " (n_rank=", _n_rank, ").");
    for (unsigned int d = 0; d < 3; ++d)
    {
      begin[d] = _begin[d][rank];
      end[d] = _end[d][rank];
    }
  }
  /// cells of this rank's real-space block
  int64_t getNumberOfCells() const { return _n_local[0] * _n_local[1] * _n_local[2]; }
  int64_t getGlobalNumberOfCells() const { return _n[0] * _n[1] * _n[2]; }

private:
  unsigned int _dim, _n_rank = 1, _rank = 0;
  std::array<int64_t, 3> _n, _n_reciprocal, _n_local, _n_reciprocal_local;
  std::array<std::vector<int64_t>, 3> _begin, _end;
  RealVectorValue _min, _max;
  std::vector<int64_t> _shape_store, _reciprocal_store;
  torch::IntArrayRef _shape, _reciprocal_shape;
};

/// include/tensor_buffers/TensorBuffer.h:17-118
struct TensorBufferStub
{
  torch::Tensor u;
  std::vector<torch::Tensor> u_old;
  std::size_t max_states = 0;
  std::size_t advanceState()
  {
    if (u_old.size() < max_states)
      u_old.resize(u_old.size() + 1);
    for (std::size_t i = u_old.size(); i-- > 1;)
      u_old[i] = u_old[i - 1];
    if (!u_old.empty())
      u_old[0] = u;
    return u_old.size();
  }
};

class TensorOperatorBase;

/// the slice of TensorProblem (+ FEProblemBase time bookkeeping) that solvers and computes reach
class TensorProblem
{
public:
  TensorProblem(const DomainAction & domain) : _domain(domain) {}
  const DomainAction & domain() const { return _domain; }
  template <typename T = torch::Tensor>
  T & getBuffer(const std::string & buffer_name, unsigned int = 0)
  {
    return slot(buffer_name).u;
  }
  template <typename T = torch::Tensor>
  const std::vector<T> & getBufferOld(const std::string & buffer_name, unsigned int max_states)
  {
    auto & b = slot(buffer_name);
    b.max_states = std::max<std::size_t>(b.max_states, max_states);
    return b.u_old;
  }
  void registerGhostLayerRequest(const std::string &, unsigned int) {}
  Real & subDt() { return _sub_dt; }
  Real & subTime() { return _sub_time; }
  Real & dt() { return _dt; }
  Real & dtOld() { return _dt_old; }
  Real & time() { return _time; }
  Real & timeOld() { return _time_old; }
  int & timeStep() { return _t_step; }
  /// src/problems/TensorProblem.C:451-472
  void advanceState()
  {
    if (timeStep() <= 1)
      return;
    for (auto & kv : _buffers)
      kv.second.advanceState();
  }
  typedef std::vector<std::shared_ptr<TensorOperatorBase>> TensorComputeList;
  const TensorComputeList & getComputes() const { return _computes; }
  TensorComputeList & computes() { return _computes; }

private:
  TensorBufferStub & slot(const std::string & name) { return _buffers[name]; }
  const DomainAction & _domain;
  std::map<std::string, TensorBufferStub> _buffers;
  TensorComputeList _computes;
  Real _sub_dt = 0, _sub_time = 0, _dt = 0, _dt_old = 0, _time = 0, _time_old = 0;
  int _t_step = 0;
};

/// include/tensor_computes/TensorOperatorBase.h:26-151
class TensorOperatorBase : public MooseObject
{
public:
  static InputParameters validParams() { return InputParameters(); }
  TensorOperatorBase(const InputParameters & parameters)
    : MooseObject(parameters),
      _tensor_problem(*parameters.get<TensorProblem *>("_tensor_problem")),
      _domain(_tensor_problem.domain()),
      _time(_tensor_problem.subTime()),
      _dim(_domain.getDim())
  {
  }
  virtual void updateDependencies() {}
  virtual void computeBuffer() = 0;
  virtual bool supportsJIT() const { return true; }
  template <typename T = torch::Tensor>
  const T & getInputBuffer(const std::string & param, unsigned int = 0)
  {
    return getInputBufferByName<T>(getParam<TensorInputBufferName>(param));
  }
  template <typename T = torch::Tensor>
  const T & getInputBufferByName(const TensorInputBufferName & buffer_name, unsigned int = 0)
  {
    _requested_buffers.insert(buffer_name);
    return _tensor_problem.getBuffer<T>(buffer_name);
  }
  template <typename T = torch::Tensor>
  T & getOutputBuffer(const std::string & param)
  {
    return getOutputBufferByName<T>(getParam<TensorOutputBufferName>(param));
  }
  template <typename T = torch::Tensor>
  T & getOutputBufferByName(const TensorOutputBufferName & buffer_name)
  {
    _supplied_buffers.insert(buffer_name);
    return _tensor_problem.getBuffer<T>(buffer_name);
  }
  std::set<std::string> _requested_buffers, _supplied_buffers;
  TensorProblem & _tensor_problem;
  const DomainAction & _domain;
  const Real & _time;
  const unsigned int & _dim;
};

/// include/tensor_computes/TensorOperator.h:17-45
template <typename T = torch::Tensor>
class TensorOperator : public TensorOperatorBase
{
public:
  static InputParameters validParams()
  {
    InputParameters params = TensorOperatorBase::validParams();
    params.addRequiredParam<TensorOutputBufferName>("buffer", "The buffer this compute is writing to");
    return params;
  }
  TensorOperator(const InputParameters & parameters) : TensorOperatorBase(parameters), _u(getOutputBuffer<T>("buffer")) {}

protected:
  T & _u;
};

/// src/tensor_computes/ComputeGroup.C:61-84 (members run in the order given: the driver lists them in dependency order)
class ComputeGroup : public TensorOperatorBase
{
public:
  ComputeGroup(const InputParameters & parameters) : TensorOperatorBase(parameters) {}
  void add(std::shared_ptr<TensorOperatorBase> c) { _members.push_back(std::move(c)); }
  virtual void computeBuffer() override
  {
    for (auto & m : _members)
      m->computeBuffer();
  }

private:
  std::vector<std::shared_ptr<TensorOperatorBase>> _members;
};

/// include/tensor_solver/TensorSolver.h:17-62, src/tensor_solver/TensorSolver.C:14-109
class TensorSolver : public TensorOperatorBase
{
public:
  static InputParameters validParams()
  {
    InputParameters params = TensorOperatorBase::validParams();
    params.addParam<TensorComputeName>("root_compute", "Primary compute object that updates the buffers");
    params.addParam<unsigned int>("substeps", 1, "Solver substeps per time step.");
    params.addParam<std::vector<TensorOutputBufferName>>("forward_buffer", {}, "Buffers updated from forward_buffer_new");
    params.addParam<std::vector<TensorInputBufferName>>("forward_buffer_new", {}, "New values to update `forward_buffer` with.");
    return params;
  }
  TensorSolver(const InputParameters & parameters)
    : TensorOperatorBase(parameters),
      _substeps(getParam<unsigned int>("substeps")),
      _substep(0),
      _sub_dt(_tensor_problem.subDt()),
      _sub_time(_tensor_problem.subTime()),
      _dt(_tensor_problem.dt()),
      _dt_old(_tensor_problem.dtOld())
  {
    const auto & to = getParam<std::vector<TensorOutputBufferName>>("forward_buffer");
    const auto & from = getParam<std::vector<TensorInputBufferName>>("forward_buffer_new");
    if (to.size() != from.size())
      paramError("forward_buffer", "needs one forward_buffer_new per entry");
    for (std::size_t i = 0; i < to.size(); ++i)
      _forwarded_buffers.emplace_back(getOutputBufferByName(to[i]), getInputBufferByName(from[i]));
  }
  virtual void computeBuffer() override
  {
    _sub_time = _time;
    _sub_dt = _dt / _substeps;
    for (_substep = 0; _substep < _substeps; _substep++)
    {
      substep();
      if (_substep < _substeps - 1)
        _tensor_problem.advanceState();
      _sub_time += _sub_dt;
    }
  }
  virtual void updateDependencies() override
  {
    if (!isParamValid("root_compute"))
      return;
    for (const auto & cmp : _tensor_problem.getComputes())
      if (cmp->name() == getParam<TensorComputeName>("root_compute"))
      {
        _compute = cmp;
        return;
      }
    paramError("root_compute", "Compute object not found.");
  }
  virtual bool supportsJIT() const override { return false; }

protected:
  const std::vector<torch::Tensor> & getBufferOldByName(const TensorInputBufferName & buffer_name, unsigned int max_states)
  {
    return _tensor_problem.getBufferOld(buffer_name, max_states);
  }
  void forwardBuffers()
  {
    for (const auto & fb : _forwarded_buffers)
      fb.first = fb.second;
  }
  virtual void substep() = 0;

  const unsigned int _substeps;
  unsigned int _substep;
  Real & _sub_dt;
  Real & _sub_time;
  const Real & _dt;
  const Real & _dt_old;
  std::shared_ptr<TensorOperatorBase> _compute;
  std::vector<std::pair<torch::Tensor &, const torch::Tensor &>> _forwarded_buffers;
};

/// include/tensor_solver/SplitOperatorBase.h:17-37, src/tensor_solver/SplitOperatorBase.C:13-64
class SplitOperatorBase : public TensorSolver
{
public:
  static InputParameters validParams()
  {
    InputParameters params = TensorSolver::validParams();
    params.addRequiredParam<std::vector<TensorOutputBufferName>>("buffer", "The buffer this solver is writing to");
    params.addRequiredParam<std::vector<TensorInputBufferName>>("reciprocal_buffer", "Reciprocal of the integrated buffer");
    params.addRequiredParam<std::vector<TensorInputBufferName>>("linear_reciprocal", "Reciprocal of the linear prefactor");
    params.addRequiredParam<std::vector<TensorInputBufferName>>("nonlinear_reciprocal", "Reciprocal of the non-linear contribution");
    return params;
  }
  SplitOperatorBase(const InputParameters & parameters) : TensorSolver(parameters) {}

protected:
  struct Variable
  {
    torch::Tensor & _buffer;
    const torch::Tensor & _reciprocal_buffer;
    const torch::Tensor * _linear_reciprocal;
    const torch::Tensor & _nonlinear_reciprocal;
    const std::vector<torch::Tensor> & _old_nonlinear_reciprocal;
  };
  void getVariables(unsigned int history_size)
  {
    const auto buffers = getParam<std::vector<TensorOutputBufferName>>("buffer");
    const auto ubar = getParam<std::vector<TensorInputBufferName>>("reciprocal_buffer");
    auto lin = getParam<std::vector<TensorInputBufferName>>("linear_reciprocal");
    const auto nonlin = getParam<std::vector<TensorInputBufferName>>("nonlinear_reciprocal");
    const auto n = buffers.size();
    if (lin.empty())
      lin.assign(n, "0");
    if (ubar.size() != n || lin.size() != n || nonlin.size() != n)
      paramError("buffer", "Must have the same number of entries as 'reciprocal_buffer', 'linear_reciprocal' and 'nonlinear_reciprocal'.");
    for (std::size_t i = 0; i < n; ++i)
      _variables.push_back(Variable{getOutputBufferByName(buffers[i]),
                                    getInputBufferByName(ubar[i]),
                                    lin[i] == "0" ? nullptr : &getInputBufferByName(lin[i]),
                                    getInputBufferByName(nonlin[i]),
                                    getBufferOldByName(nonlin[i], history_size)});
  }
  std::vector<Variable> _variables;
};

/// include/tensor_predictor/TensorPredictor.h (what IterativeTensorSolverInterface::applyPredictors calls)
class TensorPredictor
{
public:
  virtual ~TensorPredictor() = default;
  virtual void computeBuffer() = 0;
};

/// include/tensor_solver/IterativeTensorSolverInterface.h:17-33, src/tensor_solver/IterativeTensorSolverInterface.C:13-24
class IterativeTensorSolverInterface
{
public:
  IterativeTensorSolverInterface() : _iterations(0), _is_converged(true) {}
  const unsigned int & getIterations() const { return _iterations; }
  const bool & isConverged() const { return _is_converged; }
  void addPredictor(std::shared_ptr<TensorPredictor> predictor) { _predictors.push_back(std::move(predictor)); }

protected:
  void applyPredictors()
  {
    for (const auto & pred : _predictors)
      pred->computeBuffer();
  }
  unsigned int _iterations;
  bool _is_converged;
  std::vector<std::shared_ptr<TensorPredictor>> _predictors;
};

/// include/tensor_timeintegrators/TensorTimeIntegrator.h:17-33 (the legacy [TensorTimeIntegrators] base: a TensorOperator with access
/// to old buffer states and the substep size)
template <typename T = torch::Tensor>
class TensorTimeIntegrator : public TensorOperator<T>
{
public:
  static InputParameters validParams() { return TensorOperator<T>::validParams(); }
  TensorTimeIntegrator(const InputParameters & parameters) : TensorOperator<T>(parameters), _sub_dt(this->_tensor_problem.subDt()) {}

protected:
  const std::vector<T> & getBufferOld(const std::string & param, unsigned int max_states)
  {
    return getBufferOldByName(this->template getParam<TensorInputBufferName>(param), max_states);
  }
  const std::vector<T> & getBufferOldByName(const TensorInputBufferName & buffer_name, unsigned int max_states)
  {
    return this->_tensor_problem.template getBufferOld<T>(buffer_name, max_states);
  }
  const Real & _sub_dt;
};
