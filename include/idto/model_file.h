// model_file.h — reads a multibody model (idto_amd/models/*.model, written by tools/convert_models.py from the reference's
// URDF / SDF files) into the flat tables of include/idto_model.h, for C++ consumers with no Python in the process.
//
// The reference obtains the same data from Drake: `Parser(plant).AddModels(urdf_file)` in every example's
// CreatePlantModel (e.g. reference examples/acrobot/acrobot.cc:31-35), then `TrajectoryOptimizer(diagram, plant, ...)`
// (optimizer/trajectory_optimizer.cc:43-72).  Here:
//
//   idto::ModelFile mf = idto::ModelFile::Load("idto_amd/models/acrobot.model");
//   idto::optimizer::TrajectoryOptimizer<double> opt(mf.c_model(), time_step, prob, params);
//
// The counterpart of idto_amd/model.py load_model (same grammar: a token stream `key value ...`); the tree checks are
// idto_hip_create's (include/idto_model.h states what the device evaluates).  Header-only, throws std::runtime_error.
#pragma once

#include <fstream>
#include <sstream>
#include <stdexcept>
#include <string>
#include <vector>

#include "idto_model.h"

namespace idto {

class ModelFile {
 public:
  std::string name;
  std::vector<std::string> body_names;
  std::vector<int> parent, jtype, qstart, vstart, actuated, geom_body, geom_type, pair_a, pair_b, body_path, pair_path;
  std::vector<double> X_PF, axis, mass, com, inertia, damping, geom_X, geom_size;
  double gravity[3] = {0.0, 0.0, -9.81};
  int npaths = 1, common_body = -1, nq = 0, nv = 0;

  int nbodies() const { return (int)parent.size(); }
  // optimizer/trajectory_optimizer.cc:63-72: the unactuated degrees of freedom (none when the model has no actuators at all)
  std::vector<int> unactuated_dofs() const {
    int nact = 0;
    for (int a : actuated) nact += a;
    std::vector<int> out;
    if (nact == 0) return out;
    for (int i = 0; i < nv; ++i)
      if (!actuated[i]) out.push_back(i);
    return out;
  }

  // pointers into this object's vectors: valid while it lives and is not modified (the optimizer copies the tables)
  idto_model_t c_model() const {
    idto_model_t m{};
    m.nbodies = nbodies(); m.nq = nq; m.nv = nv;
    m.parent = parent.data(); m.jtype = jtype.data(); m.qstart = qstart.data(); m.vstart = vstart.data();
    m.X_PF = X_PF.data(); m.axis = axis.data(); m.mass = mass.data(); m.com = com.data(); m.inertia = inertia.data();
    m.damping = damping.data(); m.actuated = actuated.data();
    for (int i = 0; i < 3; ++i) m.gravity[i] = gravity[i];
    m.ngeoms = (int)geom_body.size(); m.geom_body = geom_body.data(); m.geom_type = geom_type.data();
    m.geom_X = geom_X.data(); m.geom_size = geom_size.data();
    m.npairs = (int)pair_a.size(); m.pair_a = pair_a.data(); m.pair_b = pair_b.data();
    m.npaths = npaths; m.common_body = common_body; m.body_path = body_path.data(); m.pair_path = pair_path.data();
    return m;
  }

  static ModelFile Load(const std::string& path) {
    std::ifstream in(path);
    if (!in) throw std::runtime_error("ModelFile: cannot open " + path);
    std::stringstream ss;
    ss << in.rdbuf();
    std::vector<std::string> tok;
    for (std::string t; ss >> t;) tok.push_back(t);
    std::size_t pos = 0;
    auto nxt = [&]() -> const std::string& {
      if (pos >= tok.size()) throw std::runtime_error("ModelFile: " + path + " ends early");
      return tok[pos++];
    };
    auto expect = [&](const char* s) {
      const std::string& t = nxt();
      if (t != s) throw std::runtime_error("ModelFile: " + path + ": expected '" + s + "', got '" + t + "'");
    };
    auto num = [&]() { return std::stod(nxt()); };
    auto integer = [&]() { return std::stoi(nxt()); };
    auto floats = [&](std::vector<double>* v, int n) { for (int i = 0; i < n; ++i) v->push_back(num()); };
    auto joint = [&](const std::string& s) {
      if (s == "revolute") return (int)IDTO_JOINT_REVOLUTE;
      if (s == "prismatic") return (int)IDTO_JOINT_PRISMATIC;
      if (s == "planar") return (int)IDTO_JOINT_PLANAR;
      if (s == "floating") return (int)IDTO_JOINT_FLOATING;
      throw std::runtime_error("ModelFile: unknown joint type " + s);
    };
    static const int kNq[4] = {1, 1, 3, 7}, kNv[4] = {1, 1, 3, 6};

    ModelFile m;
    expect("idto_model");
    if (nxt() != "1") throw std::runtime_error("ModelFile: " + path + ": format version 1 expected");
    expect("name"); m.name = nxt();
    expect("gravity"); for (double& g : m.gravity) g = num();
    expect("nbodies"); const int nb = integer();
    expect("npaths"); m.npaths = integer();
    expect("common_body"); m.common_body = integer();
    for (int i = 0; i < nb; ++i) {
      expect("body");
      if (integer() != i) throw std::runtime_error("ModelFile: " + path + ": bodies out of order");
      m.body_names.push_back(nxt());
      expect("parent"); m.parent.push_back(integer());
      expect("joint"); m.jtype.push_back(joint(nxt()));
      expect("path"); m.body_path.push_back(integer());
      expect("X_PF"); floats(&m.X_PF, 12);
      expect("axis"); floats(&m.axis, 3);
      expect("mass"); floats(&m.mass, 1);
      expect("com"); floats(&m.com, 3);
      expect("inertia"); floats(&m.inertia, 6);
      if (m.parent.back() >= i) throw std::runtime_error("ModelFile: " + path + ": bodies must be topologically ordered");
      m.qstart.push_back(m.nq); m.vstart.push_back(m.nv);
      m.nq += kNq[m.jtype.back()]; m.nv += kNv[m.jtype.back()];
    }
    expect("damping"); floats(&m.damping, m.nv);
    expect("actuated"); for (int i = 0; i < m.nv; ++i) m.actuated.push_back(integer());
    expect("ngeoms"); const int ng = integer();
    for (int g = 0; g < ng; ++g) {
      expect("geom");
      if (integer() != g) throw std::runtime_error("ModelFile: " + path + ": geometries out of order");
      expect("body"); m.geom_body.push_back(integer());
      expect("type");
      const std::string& ty = nxt();
      if (ty != "sphere" && ty != "box") throw std::runtime_error("ModelFile: unknown geometry type " + ty);
      m.geom_type.push_back(ty == "sphere" ? (int)IDTO_GEOM_SPHERE : (int)IDTO_GEOM_BOX);
      expect("size"); floats(&m.geom_size, 3);
      expect("X_BG"); floats(&m.geom_X, 12);
    }
    expect("npairs"); const int np = integer();
    for (int k = 0; k < np; ++k) {
      expect("pair"); m.pair_a.push_back(integer()); m.pair_b.push_back(integer());
      expect("path"); m.pair_path.push_back(integer());
    }
    return m;
  }
};

}  // namespace idto
